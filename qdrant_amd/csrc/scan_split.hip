// scan_split.hip — f32 dot / cosine brute-force top-k for 65..128 queries per pass: a matrix-core PREFILTER with exact verification.
//
// Why.  The exact chain-major scan (scan_mfma16.hip) reproduces dot_similarity_avx bit for bit on v_mfma_f32_16x16x4_f32, which runs
// at the vector-ALU rate (155 TFLOP/s measured, profiles/r2_mfma_issue_rates.txt): 64 queries x 10 M x 768 cost 6.3 ms of matrix time at
// peak against a 4.5 ms HBM stream, and every further query costs another 0.1 ms.  The f16 matrix instruction is 16x faster.  An f32 value
// splits EXACTLY into two f16 values plus a residual below 2^-22 of it (x 2^e = h + l + r, h = f16(x 2^e), l = f16(x 2^e - h)), so
//     sum x_i y_i  =  2^-(ex + ey) [ sum h h' + sum h l' + sum l h' ]  +  O(2^-21) sum |x_i y_i|
// three v_mfma_f32_16x16x32_f16 per 16 x 16 x 32 block, f32 accumulation.  That approximate score is NOT returned to anybody: it only
// decides which rows are worth an exact look.
//   1. prescan   (api_search.hip)  exact scores of a strided sample of the block -> T_q = exact k-th best of the sample <= final k-th best
//   2. this file            approximate score A(r, q) of EVERY row; rows with A >= T_q - b_q become candidates (~1000 k per query)
//   3. select               A_k = k-th best approximate score among the candidates; keep those with A >= A_k - 2 b_q  (~k rows)
//   4. verify    (api_search.hip)  exact scores of the kept rows with the gather kernel of qmx_rescore (the reference's bits), sort, top k
// With |A - E| <= b_q for the exact score E (b_q = 1e-4 |q| max|row|, two orders above the split error and above the worst-case f32
// accumulation bound of both sides for dim <= 1600) every member of the exact top k survives 2. and 3., so the result is the exact
// scan's result, bit for bit — ids, scores, ties — and the parity tests run against this path unchanged.  Anything unexpected (a
// candidate buffer or a verification list that overflows: masses of equal scores, a sample that is all deleted) raises a device-side
// flag and the exact scan runs after all in the same stream (its kernels start, read the flag and return when it is clear).
//
// The kernel is a GEMM with the stored block as the streamed operand: C[rows x queries] over K = dim.
//   block = 512 threads = 8 waves = 4 (row quarters) x 2 (query halves), one block per CU, persistent over 256-row tiles.
//   per K-chunk of 32 floats: the 256 x 128-byte row pieces come from HBM with the scan's access pattern (8 lanes per row, one full
//   128-byte line per row and instruction, every byte once), are split into h / l on the fly and land in LDS in the A-operand layout of
//   the instruction (16-byte units: 8 consecutive k of one row); the queries were split once per batch into the same layout in global
//   memory (393 KB at 128 x 768, L2-resident) and their chunk is copied to LDS next to the rows.  Double-buffered, one barrier per chunk.
//   A wave multiplies its 64 rows x 64 queries: 16 accumulator tiles (64 VGPRs), 16 ds_read_b128 and 48 MFMAs per chunk.
//   LDS units are swizzled (row ^ 2 kq) so that both the 8-byte stores of the loaders and the 16-byte loads of the MFMA lanes are
//   conflict-free in the bank groups of /opt/skills/guides/MI355X_MICROARCH.md (LDS table).
// Roofline: HBM (3072 B per row at d = 768, once); matrix time 3 x 2 x 128 x 768 flop per row = 2.4 ms per 10 M rows at the f16 peak.
#include <type_traits>

#include "scan_common.hpp"

namespace qmx {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4s __attribute__((ext_vector_type(4)));

constexpr int SP_BM = 256;            // rows per tile
constexpr int SP_QT = 128;            // queries per pass
constexpr int SP_QT_MAX = 256;        // ... of the 256-query shape over the half copy (scan_f16half256_kernel)
constexpr int SP_THREADS = 768;          // 8 consumer waves (matrix cores) + 4 producer waves (HBM stream, f32 -> f16 pairs): 2 + 1 per SIMD
constexpr int SP_CONSUMERS = 8;
constexpr int SP_A_UNITS = SP_BM * 2 * 4;     // 16-byte units of one A chunk buffer (256 rows x {h, l} x 4 k-groups) = 32 KB
constexpr int SP_B_UNITS = SP_QT * 2 * 4;     // ... of one B chunk buffer = 16 KB
constexpr int SP_BRING = 4;           // LDS buffers of the queries' chunks: an LDS-DMA copy lands ~1.1 us after its issue (MI355X guide, ldsdma-fill),
                                      // longer than a stage lasts, so chunk g + 3 is requested while chunk g is multiplied
constexpr int SP_LDS = (2 * SP_A_UNITS + SP_BRING * SP_B_UNITS) * 16;

// unit index of (16-row or 16-query tile t, half hl, k-group kq, row-in-tile m) inside a chunk buffer
__device__ __forceinline__ uint32_t sp_unit(uint32_t t, uint32_t hl, uint32_t kq, uint32_t m) { return ((t * 2 + hl) * 4 + kq) * 16 + (m ^ (2 * kq)); }

// x * scale = h + l (+ a residual below 2^-22 |x * scale|); round to nearest even both times
// (scale is a power of two, so x * scale is exact and fma(x, scale, -h) == x * scale - h: two v_fma_mix instructions per value)
__device__ __forceinline__ void sp_split4(const f32x4s v, float scale, half4 &h, half4 &l) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = (_Float16)__builtin_fmaf(v[i], scale, 0.0f);
        l[i] = (_Float16)__builtin_fmaf(v[i], scale, -(float)h[i]);
    }
}

// ---- once per query batch: preprocessed f32 queries -> split f16 in the B-operand layout, per-query norms, the scale of the batch ----
// stats[0] = max |q| over the batch (uint bits of a non-negative float order like the float)
__global__ void sp_query_stats_kernel(const float *q, uint32_t nq, uint32_t dim, float *qmax, float *qnorm) {
    const uint32_t qi = blockIdx.x;
    float mx = 0.0f, ss = 0.0f;
    for (uint32_t i = threadIdx.x; i < dim; i += blockDim.x) {
        const float v = q[(uint64_t)qi * dim + i];
        mx = __builtin_fmaxf(mx, __builtin_fabsf(v));
        ss = __builtin_fmaf(v, v, ss);
    }
    __shared__ float smx[256], sss[256];
    smx[threadIdx.x] = mx;
    sss[threadIdx.x] = ss;
    __syncthreads();
    for (uint32_t o = blockDim.x / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            smx[threadIdx.x] = __builtin_fmaxf(smx[threadIdx.x], smx[threadIdx.x + o]);
            sss[threadIdx.x] += sss[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        qmax[qi] = smx[0];
        qnorm[qi] = __builtin_sqrtf(sss[0]);
    }
}
// power of two that brings a magnitude bound to [8192, 16384): headroom below the f16 maximum, the low part stays a normal number
__device__ __forceinline__ float sp_pow2_scale(float maxabs) {
    if (!(maxabs > 0.0f) || !(maxabs < 3.0e38f)) return 1.0f;
    int e;
    (void)__builtin_frexpf(maxabs, &e);          // maxabs = m * 2^e, m in [0.5, 1)
    int s = 14 - e;
    s = s > 100 ? 100 : (s < -100 ? -100 : s);
    return __builtin_ldexpf(1.0f, s);
}
// scales[0] = query scale, scales[1] = row scale * query scale (accumulator units per score unit), scales[2] = its inverse
// the batch's query scale from the per-query maxima (at most 256 of them), by every thread of a block of a multiple of 64 threads (<= 256); `sh`: 4 floats of LDS
__device__ __forceinline__ float sp_batch_scale(const float *qmax, uint32_t nq, float *sh) {
    float mx = 0.0f;
    for (uint32_t i = threadIdx.x; i < nq; i += blockDim.x) mx = __builtin_fmaxf(mx, qmax[i]);
    for (int o = 32; o >= 1; o >>= 1) mx = __builtin_fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = sh[0];
    for (uint32_t w = 1; w < blockDim.x / 64; ++w) mx = __builtin_fmaxf(mx, sh[w]);
    return sp_pow2_scale(mx);
}
// one thread per 16-byte unit: bq[kc][nt][hl][kq][n ^ 2 kq] = 8 halfs, k = 32 kc + 8 kq + e, query 16 nt + n (zero beyond nq)
// half != 0 (the one-product mode): a chunk is 64 floats, the two unit planes hold the high parts of its two 32-float halves
__global__ __launch_bounds__(256) void sp_pack_queries_kernel(const float *q, uint32_t nq, uint32_t dim, const float *qmax, uint4 *bq, int half, uint32_t qt) {
    __shared__ float sh_mx[4];
    const float scale = sp_batch_scale(qmax, nq, sh_mx);
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (dim / 32) * qt * 4) return;
    const uint32_t b_units = qt * 8;                                    // 16-byte units of one chunk of the tile (128 queries: SP_B_UNITS)
    const uint32_t k32 = gid / (qt * 4), r = gid % (qt * 4);            // 32-float group of the row
    const uint32_t nt = r / 64, kq = (r / 16) % 4, n = r % 16;
    const uint32_t qi = nt * 16 + n;
    half8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = qi < nq ? q[(uint64_t)qi * dim + k32 * 32 + kq * 8 + e] * scale : 0.0f;
        h[e] = (_Float16)x;
        l[e] = (_Float16)(x - (float)h[e]);
    }
    if (half) {
        bq[(uint64_t)(k32 / 2) * b_units + sp_unit(nt, k32 & 1u, kq, n)] = *reinterpret_cast<const uint4 *>(&h);
        return;
    }
    uint4 *chunk = bq + (uint64_t)k32 * b_units;
    chunk[sp_unit(nt, 0, kq, n)] = *reinterpret_cast<const uint4 *>(&h);
    chunk[sp_unit(nt, 1, kq, n)] = *reinterpret_cast<const uint4 *>(&l);
}
// thr[q] in accumulator units: (exact k-th best of the sample - band) * scale.  band[q] = rel_band * row_norm_max * |q| (score units).
// Also: the batch's scales (scales[0] = query scale, [1] = row scale x query scale = accumulator units per score unit, [2] = its inverse) and the
// zeroed candidate counters of the pass.
__global__ __launch_bounds__(256) void sp_thresholds_kernel(const uint64_t *gthr, const float *qnorm, const float *qmax, uint32_t nq, float rel_band,
                                                            float row_norm_max, float row_scale, float *scales, float *thr, float *band, uint32_t *cand_cnt,
                                                            uint32_t n_cnt) {
    __shared__ float sh_mx[4];
    const float qs = sp_batch_scale(qmax, nq, sh_mx);
    const uint32_t q = threadIdx.x;                   // blockDim.x = queries of the tile (128 | 256)
    if (q == 0) {
        scales[0] = qs;
        scales[1] = row_scale * qs;
        scales[2] = 1.0f / (row_scale * qs);
    }
    for (uint32_t i = q; i < n_cnt; i += blockDim.x) cand_cnt[i] = 0;
    if (q >= nq) { thr[q] = __builtin_inff(); band[q] = 0.0f; return; }
    const float b = rel_band * row_norm_max * qnorm[q];
    const uint64_t k = gthr[q];
    // no bound (the sample holds fewer than k live rows): no candidates for this query, and its infinite band sends it - alone - to the exact scan
    band[q] = k ? b : __builtin_inff();
    thr[q] = k ? (key_score(k) - b) * (row_scale * qs) : __builtin_inff();
}

typedef __attribute__((address_space(3))) unsigned char sp_lds_byte;
// 1 KiB of global memory (wave-uniform base, lane i fetches bytes 16 i .. 16 i + 15) straight into LDS at the wave-uniform byte address
// lds_dst (lane i lands at lds_dst + 16 i): no staging registers.  The compiler does not count these loads: the kernel waits for them
// itself (s_waitcnt vmcnt(0) in front of the chunk barrier).
__device__ __forceinline__ void sp_glds16(const unsigned char *src, uint32_t lane_off, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(src), "s"(lds_dst) : "memory");
}

// the same for bytes that one CU reads once (the rows of a copy): non-temporal
__device__ __forceinline__ void sp_glds16_nt(const unsigned char *src, uint32_t lane_off, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(src), "s"(lds_dst) : "memory");
}

// the same without saving m0 (the kernel that uses it declares m0 clobbered: no other user of m0 in it)
__device__ __forceinline__ void sp_glds16_m0(const unsigned char *src, uint32_t lane_off, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane_off), "s"(src), "s"(lds_dst) : "memory");   // (m0 is a reserved register: the compiler neither allocates nor tracks it)
}

__device__ __forceinline__ void sp_glds16_m0_nt(const unsigned char *src, uint32_t lane_off, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" : : "v"(lane_off), "s"(src), "s"(lds_dst) : "memory");
}

struct SplitArgs {
    const uint4 *bq;        // split queries, [dim / 32][SP_B_UNITS]
    uint32_t nchunks;       // dim / 32
    uint32_t nq;            // live queries (<= 128)
    float row_scale;        // power of two applied to the rows before the split
    const float *scales;    // device: [1] = accumulator units per score unit, [2] = inverse
    const float *thr;       // [128] candidate threshold, accumulator units
    const uint4 *rows_split; // the pre-split copy of the block (scan_f16pair_kernel), or nullptr
    uint32_t phase;         // scan_f16pair_kernel: 0 = every tile, 1 = tiles 0, 16, 32, ... (the strided sixteenth), 2 = all the others
    uint4 *wlist;           // scan_f16pair_kernel: [waves][wcap] (key lo, key hi, query, 0): candidates in the order each wave met them
    uint32_t *wcnt;         // [waves] entries each wave wanted to append (may run past wcap: overflow)
    uint32_t wcap;
    uint64_t *cand;         // [128][cap] keys (approximate score, row)
    uint32_t *cand_cnt;     // [128] appended (may run past cap: overflow)
    uint32_t cap;
};

// The stage barrier.  NOT __syncthreads(): its workgroup fence makes the compiler wait for every outstanding global load (vmcnt(0)) in front of
// the barrier, which would drain the producers' row stream at every stage (measured: a stage then lasts one loaded HBM round trip, 2.3 us).
// LDS traffic is all that crosses this barrier: wait for this wave's LDS operations, then s_barrier.
__device__ __forceinline__ void sp_stage_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(SP_THREADS, 1) void scan_f32_split_kernel(const ScanArgs a, const SplitArgs s) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4 *lds = reinterpret_cast<uint4 *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned char *rows = reinterpret_cast<const unsigned char *>(a.rows);
    const uint64_t n_tiles = (a.n_cand + SP_BM - 1) / SP_BM;
    const uint32_t nch = s.nchunks;                       // a multiple of 4 (the host takes this path for dim % 128 == 0)
    const uint64_t my_tiles = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const uint64_t G = my_tiles * nch;                    // stages of this block = its (tile, K-chunk) pairs, in order
    if (G == 0) return;

    if (w >= SP_CONSUMERS) {
        // ================= producers (4 waves): HBM -> registers -> split -> LDS, one K-chunk per stage, three chunks of rows in flight ==========
        // rows 64 pw + 8 rr + (lane >> 3), rr = 0..7, 16-byte piece p = lane & 7 of the chunk's 128 bytes: a wave instruction reads 8 rows x one
        // full 128-byte line, every byte once
        const uint32_t pw = (uint32_t)w - SP_CONSUMERS;
        const uint32_t p = (uint32_t)lane & 7u, lrow0 = pw * 64u + ((uint32_t)lane >> 3);
        const float rscale = s.row_scale;
        // byte offset of this thread's 8-byte stores inside an A buffer: row rl = lrow0 + 8 rr -> tile rl >> 4 = 4 pw + (rr >> 1), row-in-tile
        // (lane >> 3) + 8 (rr & 1); h half (l is + 64 units)
        const uint32_t kq_w = p >> 1;
        uint32_t a_wr[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) a_wr[e] = sp_unit(pw * 4u, 0, kq_w, ((uint32_t)lane >> 3) + 8u * (uint32_t)e) * 16u + (p & 1u) * 8u;
        f32x4s areg[3][8];                                // [slot = chunk % 3][rr]
        // the tile's first row through SGPRs + one 32-bit lane offset per row group: no 64-bit vector address arithmetic per load
        typedef const __attribute__((address_space(1))) unsigned char *sp_gptr;   // (an explicit global pointer: rebuilt from SGPR halves, it would be "flat")
        sp_gptr tile_base = (sp_gptr)(uintptr_t)rows;     // wave-uniform: row 0 of the tile being requested
        uint32_t voff[8];                                 // lane part: (row-in-tile) * stride + 16 p, clamped to the rows that exist
        uint64_t ld_g = 0, ld_it = 0;                     // next chunk to request: global index, its tile iteration
        uint32_t ld_kc = 0;
        auto set_tile = [&](uint64_t it) {
            const uint64_t tile = blockIdx.x + it * gridDim.x;
            const uint64_t first = tile * SP_BM;
            const uint64_t left = a.n_cand - first;       // rows of the tile that exist (> 0: the tile is one of this block's)
            const uint64_t tb = (uint64_t)(uintptr_t)(rows + first * a.row_stride);
            tile_base = (sp_gptr)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(tb >> 32)) << 32) |
                                             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tb));
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {              // rows past the end of the block re-read its last row: results masked
                const uint64_t rl = lrow0 + 8u * (uint32_t)rr;
                voff[rr] = (uint32_t)((rl < left ? rl : left - 1) * a.row_stride) + p * 16u;
            }
        };
        auto issue = [&](int slot) {                      // request the next chunk into register slot `slot`
            if (ld_g >= G) return;
            sp_gptr cb = tile_base + ld_kc * 128u;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr)
                areg[slot][rr] = __builtin_nontemporal_load(reinterpret_cast<const __attribute__((address_space(1))) f32x4s *>(cb + voff[rr]));
            ++ld_g;
            if (++ld_kc == nch) {
                ld_kc = 0;
                ++ld_it;
                if (ld_g < G) set_tile(ld_it);
            }
        };
        auto convert = [&](int slot, uint32_t buf) {      // slot -> A buffer `buf`
            unsigned char *ab = reinterpret_cast<unsigned char *>(lds + buf * SP_A_UNITS);
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                half4 h, l;
                sp_split4(areg[slot][rr], rscale, h, l);
                const uint32_t off = a_wr[rr & 1] + (uint32_t)(rr >> 1) * (128u * 16u);   // + one 16-row tile (128 units) per two rr
                *reinterpret_cast<half4 *>(ab + off) = h;
                *reinterpret_cast<half4 *>(ab + off + 64 * 16) = l;
            }
        };
        set_tile(0);
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) issue(sl);
        convert(0, 0);
        issue(0);
        sp_stage_barrier();                                  // stage 0 may start
        for (uint64_t g = 0; g < G; g += 6) {             // (unrolled by 6: the register slot is g % 3, the LDS buffer g % 2)
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                if (g + u >= G) break;
                // stage g + u: the consumers multiply buffer (g + u) & 1 while chunk g + u + 1 is prepared in the other one
                if (g + u + 1 < G) {
                    convert((u + 1) % 3, (uint32_t)(u + 1) & 1u);
                    issue((u + 1) % 3);
                }
                sp_stage_barrier();
            }
        }
        return;
    }

    // ================= consumers (8 waves = 4 row quarters x 2 query halves): LDS -> matrix cores, candidates of every finished tile ==========
    const uint32_t wm = (uint32_t)w & 3u, wn = (uint32_t)w >> 2;
    const uint32_t kq_r = (uint32_t)lane >> 4, m_r = (uint32_t)lane & 15u;
    const uint32_t a_rd = sp_unit(wm * 4, 0, kq_r, m_r), b_rd = sp_unit(wn * 4, 0, kq_r, m_r);   // + 128 units per tile, + 64 for the l half
    float thr[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) thr[nt] = s.thr[wn * 64 + nt * 16 + m_r];
    const float inv_scale = s.scales[2];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(sp_lds_byte *)smem_raw;
    const uint32_t lane_off = (uint32_t)lane * 16u;
    // the queries' chunk kc -> LDS buffer `bbuf` directly (this wave's 2 KiB of its 16 KiB): the only vector-memory traffic of a consumer
    auto load_b = [&](uint32_t kc, uint32_t bbuf) {
        const unsigned char *bsrc = reinterpret_cast<const unsigned char *>(s.bq + (uint64_t)kc * SP_B_UNITS) + (uint32_t)w * 2048u;
        const uint32_t dst = lds0 + (2u * SP_A_UNITS + bbuf * SP_B_UNITS) * 16u + (uint32_t)w * 2048u;
        // (wave-uniform base through SGPRs; readfirstlane returns a signed int: widen the halves as unsigned)
        const uint64_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)bsrc);
        const uint64_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uintptr_t)bsrc >> 32));
        const unsigned char *ub = reinterpret_cast<const unsigned char *>((hi << 32) | lo);
        sp_glds16(ub, lane_off, dst);
        sp_glds16(ub + 1024u, lane_off, dst + 1024u);
    };
    // chunks 0, 1, 2 are on their way before stage 0; every stage requests one more (indices wrap: the queries are the same for every tile;
    // the requests past the last stage land in buffers nobody reads) so that the count of outstanding copies is the same at every wait
    uint32_t b_kc = 0;                                    // K-chunk of the next request
    uint32_t b_slot = 0;                                  // ... and its ring slot
    auto request_b = [&]() {
        load_b(b_kc, b_slot);
        b_kc = b_kc + 1 == nch ? 0 : b_kc + 1;
        b_slot = (b_slot + 1) & (SP_BRING - 1);
    };
    request_b();
    request_b();
    request_b();
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // chunk 0 has landed (2 copies per chunk and wave; chunks 1 and 2 may still travel)
    sp_stage_barrier();
    uint32_t buf = 0, bbuf = 0;
    for (uint64_t it = 0; it < my_tiles; ++it) {
        const uint64_t tile = blockIdx.x + it * gridDim.x;
        f32x4s acc[4][4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4s){0.f, 0.f, 0.f, 0.f};
        for (uint32_t kc = 0; kc < nch; ++kc) {
            request_b();                                  // chunk g + 3 -> the slot chunk g - 1 was read from (everybody is past that stage's barrier)
            const uint4 *ab = lds + buf * SP_A_UNITS + a_rd;
            const uint4 *bb = lds + 2 * SP_A_UNITS + bbuf * SP_B_UNITS + b_rd;
            half8 bh[4], bl[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                bh[nt] = *reinterpret_cast<const half8 *>(bb + nt * 128);
                bl[nt] = *reinterpret_cast<const half8 *>(bb + nt * 128 + 64);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const half8 ah = *reinterpret_cast<const half8 *>(ab + mt * 128);
                const half8 al = *reinterpret_cast<const half8 *>(ab + mt * 128 + 64);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[nt], acc[mt][nt], 0, 0, 0);
            }
            buf ^= 1;
            bbuf = (bbuf + 1) & (SP_BRING - 1);
            if (kc + 1 == nch) {
                // ---- candidates of the tile: accumulator register j of (mt, nt) = row 64 wm + 16 mt + 4 (lane >> 4) + j, query 64 wn + 16 nt + (lane & 15)
                const uint32_t row0 = (uint32_t)(tile * SP_BM) + wm * 64 + 4 * kq_r;      // (row ids are u32: PointOffsetType)
                const uint32_t n_rows32 = (uint32_t)a.n_cand;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float v = acc[mt][nt][j];
                            const uint32_t row = row0 + (uint32_t)mt * 16 + (uint32_t)j;
                            const bool c = !(v < thr[nt]) && row < n_rows32;           // NaN (greatest in OrderedFloat) is a candidate; thr = +inf past nq
                            if (__ballot(c)) {
                                uint32_t q = wn * 64 + (uint32_t)nt * 16 + m_r;
                                asm volatile("" : "+v"(q));                            // (keeps the buffer addresses out of the registers of the main loop)
                                if (c && q < s.nq && a.del.live(row)) {
                                    const uint32_t slot = atomicAdd(&s.cand_cnt[q], 1u);
                                    if (slot < s.cap) s.cand[(uint64_t)q * s.cap + slot] = make_key(v * inv_scale, row);
                                }
                            }
                        }
                    }
            }
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // the queries' NEXT chunk has landed (the two behind it may still travel)
            sp_stage_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // nothing may land in LDS after the block is gone
}

// =====================================================================================================================================
// The same scan over a PRE-SPLIT copy of the block (QMX_SEG_SPLIT_COPY): the f16 pairs are made once, when the segment is created, and laid
// out in HBM as the LDS images the multiplication wants — [256-row tile][K-chunk][2048 16-byte units] — so a stage's 32 KiB of rows are ONE
// contiguous run that the waves copy straight into LDS (LDS-DMA, 1 KiB per wave instruction, no registers, no vector-ALU work at all).
// The vector ALU shares its issue port with the matrix instructions (measured on the converting kernel above: 33 % VALU + 36 % MFMA busy,
// not overlapped), so taking the conversion out of the scan is what lets the pass run at the HBM stream.  Costs a second copy of the
// block in HBM (4 bytes per element, like the f32 original, which stays: the verification gathers from it).
// block = 512 threads = 8 waves (4 row quarters x 2 query halves), every wave multiplies and copies; three-deep rings for rows and queries
// (a copy lands ~1.1 us after its issue, a stage lasts ~0.7 us): stage g requests stage g + 2 and multiplies stage g.
// =====================================================================================================================================
constexpr int SP3_THREADS = 512;
constexpr uint32_t SP_PHASE_STRIDE = 16;
// `phase` as the launchers take it: low byte 0 = every tile, 1 = tiles 0, S, 2 S, ... (the strided sample), 2 = all the others; the bytes above = S (0: 16)
__host__ __device__ inline uint32_t split_phase_stride(uint32_t phase) { const uint32_t st = phase >> 8; return st >= 2 ? st : SP_PHASE_STRIDE; }
__host__ __device__ inline uint64_t split_phase_tiles(uint64_t all_tiles, uint32_t phase) {
    const uint32_t st = split_phase_stride(phase), ph = phase & 0xFFu;
    const uint64_t first = (all_tiles + st - 1) / st;
    return ph == 0 ? all_tiles : ph == 1 ? first : all_tiles - first;
}
constexpr uint32_t SP_WCAP = 8192;                           // candidates one wave may list per pass (expected: ~200, heavy-tailed rows under the int8 band: thousands; more -> overflow -> the exact scan)
constexpr int SP3_BM = SP_BM;                                // 256 rows per tile: a stage is 32 KiB of rows + 16 KiB of queries
constexpr int SP3_A_UNITS = SP_A_UNITS;
constexpr int SP3_ARING = 3;                                 // row stages in LDS: one being multiplied, two on their way
constexpr int SP3_BRING = 4;                                 // query stages: one being multiplied, three requested
constexpr int SP3_LDS = (SP3_ARING * SP3_A_UNITS + SP3_BRING * SP_B_UNITS) * 16;      // 96 + 64 = 160 KiB: the whole LDS of the CU

template <bool HALF /* one product per element (high parts only, 64-float chunks) instead of three */>
__global__ __launch_bounds__(SP3_THREADS, 1) void scan_f16pair_kernel(const ScanArgs a, const SplitArgs s) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4 *lds = reinterpret_cast<uint4 *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t all_tiles = (a.n_cand + SP3_BM - 1) / SP3_BM;
    const uint64_t n_tiles = split_phase_tiles(all_tiles, s.phase);
    const uint32_t nch = s.nchunks;
    const uint64_t my_tiles = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const uint32_t phase = s.phase & 0xFFu, pstride = split_phase_stride(s.phase);
    // the j-th tile of this launch: every tile | the strided sixteenth (0, 16, 32, ...: a sample that sees the whole block, whatever its
    // order) | the complement (j + j / 15 + 1 skips the multiples of 16)
    auto tile_of = [&](uint64_t j) -> uint64_t { return phase == 0 ? j : phase == 1 ? j * pstride : j + j / (pstride - 1) + 1; };
    if (my_tiles == 0) {
        if (lane == 0) s.wcnt[blockIdx.x * (SP3_THREADS / 64) + (uint32_t)w] = 0;
        return;
    }
    const uint32_t wm = (uint32_t)w & 3u, wn = (uint32_t)w >> 2;
    const uint32_t kq_r = (uint32_t)lane >> 4, m_r = (uint32_t)lane & 15u;
    const uint32_t a_rd = sp_unit(wm * 4, 0, kq_r, m_r), b_rd = sp_unit(wn * 4, 0, kq_r, m_r);
    float thr[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) thr[nt] = s.thr[wn * 64 + nt * 16 + m_r];
    const float inv_scale = s.scales[2];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // from here on the kernel counts its vector-memory traffic itself
    const uint32_t lds0 = (uint32_t)(uintptr_t)(sp_lds_byte *)smem_raw;
    const uint32_t lane_off = (uint32_t)lane * 16u;
    uint4 *b_lds = lds + SP3_ARING * SP3_A_UNITS;
    // The copy streams (LDS-DMA, 1 KiB per wave instruction): stage = (tile iteration, K-chunk) in order; per stage a wave copies 4 KiB of the
    // rows (from HBM) and 2 KiB of the queries (from L2).  The vector-memory counter retires in order, so the order of the requests fixes what a
    // wait can mean: at stage g a wave requests [queries of stage g + 3, rows of stage g + 2] and waits until only those 6 copies are out -
    // then the rows of stage g + 1 (requested a stage ago) and the queries of stage g + 2 (a stage ago, BEFORE those rows) have landed: every
    // copy has two stages to arrive.  A copy requested and awaited inside one stage makes the stage last a memory round trip (1.1 - 2 us
    // measured, whatever its size).  Requests past the last stage repeat the last one into a slot nobody reads: the count in flight is constant.
    uint64_t ra_it = 0;
    uint32_t ra_kc = 0, ra_slot = 0, rb_kc = 0, rb_slot = 0;
    auto uniform_ptr = [&](uint64_t v) {
        return reinterpret_cast<const unsigned char *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
                                                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v));
    };
    // one 1 KiB copy each, so that the requests of a stage can sit BETWEEN its matrix instructions (a wave issues in order: six copies in a
    // row in front of the MFMAs keep the matrix pipe idle for ~600 cycles per stage while both waves of the SIMD are busy requesting)
    const unsigned char *ra_src = nullptr, *rb_src = nullptr;
    uint32_t ra_dst = 0, rb_dst = 0;
    auto rows_begin = [&]() {
        const uint64_t tile = tile_of(blockIdx.x + ra_it * gridDim.x);
        ra_src = uniform_ptr((uint64_t)(uintptr_t)(s.rows_split + (tile * nch + ra_kc) * SP3_A_UNITS) + (uint32_t)w * 4096u);
        ra_dst = lds0 + (ra_slot * SP3_A_UNITS) * 16u + (uint32_t)w * 4096u;
        ra_slot = ra_slot + 1 == SP3_ARING ? 0 : ra_slot + 1;
        if (ra_kc + 1 < nch) ++ra_kc;
        else if (ra_it + 1 < my_tiles) { ra_kc = 0; ++ra_it; }
    };
    auto rows_piece = [&](int i) { sp_glds16_nt(ra_src + i * 1024, lane_off, ra_dst + i * 1024); };      // (non-temporal: the copy's lines are read once, whole, by this CU)
    auto queries_begin = [&]() {
        rb_src = uniform_ptr((uint64_t)(uintptr_t)(s.bq + (uint64_t)rb_kc * SP_B_UNITS) + (uint32_t)w * 2048u);
        rb_dst = lds0 + (SP3_ARING * SP3_A_UNITS + rb_slot * SP_B_UNITS) * 16u + (uint32_t)w * 2048u;
        rb_slot = (rb_slot + 1) & (SP3_BRING - 1);
        rb_kc = rb_kc + 1 == nch ? 0 : rb_kc + 1;
    };
    auto queries_piece = [&](int i) { sp_glds16(rb_src + i * 1024, lane_off, rb_dst + i * 1024); };
    auto request_rows = [&]() {
        rows_begin();
#pragma unroll
        for (int i = 0; i < 4; ++i) rows_piece(i);
    };
    auto request_queries = [&]() {
        queries_begin();
        queries_piece(0);
        queries_piece(1);
    };
    request_queries();                                    // queries of stage 0
    request_queries();                                    // ... 1
    request_rows();                                       // rows of stage 0
    request_queries();                                    // queries of stage 2
    request_rows();                                       // rows of stage 1
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // queries 0, 1 and rows 0 have landed
    sp_stage_barrier();
    uint32_t slot = 0, bslot = 0;
    f32x4s acc[4][4];
    // Candidates go to a list of this wave's own (position = a wave-uniform counter + the lane's rank in the ballot): plain stores, nothing
    // that RETURNS - a returning operation (an atomic slot, a deleted-flag lookup) can only be awaited together with every copy requested
    // before it (the counter retires in order), i.e. it drains the two-stage prefetch: ~2 us per tile with a hit, 0.4 ms per pass at 128
    // queries.  The tile's epilogue therefore also runs at the TOP of the next stage, in front of that stage's requests: the wait at the end
    // of a stage leaves exactly the stage's own six copies out, stores included or not.  sp_regroup_kernel sorts the lists by query afterwards
    // (and drops deleted rows).
    uint4 *const wl = s.wlist + (uint64_t)(blockIdx.x * (SP3_THREADS / 64) + (uint32_t)w) * s.wcap;
    uint32_t wcount = 0;
    auto epilogue = [&](uint64_t tile) {
        const uint32_t row0 = (uint32_t)(tile * SP3_BM) + wm * 64 + 4 * kq_r;
        const uint32_t n_rows32 = (uint32_t)a.n_cand;
        // most (wave, tile) pairs hold no candidate once the thresholds bite: one maximum per query tile decides that in 60 instructions
        bool maybe = false;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            float mx = -__builtin_inff();
            bool nan = false;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    mx = __builtin_fmaxf(mx, acc[mt][nt][j]);
                    nan = nan || acc[mt][nt][j] != acc[mt][nt][j];
                }
            maybe = maybe || !(mx < thr[nt]) || nan;
        }
        if (!__ballot(maybe)) return;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const uint32_t q = wn * 64 + (uint32_t)nt * 16 + m_r;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = acc[mt][nt][j];
                    const uint32_t row = row0 + (uint32_t)mt * 16 + (uint32_t)j;
                    const bool c = !(v < thr[nt]) && row < n_rows32 && q < s.nq;       // NaN (greatest in OrderedFloat) is a candidate
                    const uint64_t hits = __ballot(c);
                    if (hits) {
                        const uint32_t at = wcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(hits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hits, 0u));
                        if (c && at < s.wcap) {
                            const uint64_t key = make_key(v * inv_scale, row);
                            wl[at] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), q, 0u);
                        }
                        wcount += (uint32_t)__builtin_popcountll(hits);
                    }
                }
            }
    };
    for (uint64_t it = 0; it < my_tiles; ++it) {
        if (it) epilogue(tile_of(blockIdx.x + (it - 1) * gridDim.x));
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4s){0.f, 0.f, 0.f, 0.f};
        for (uint32_t kc = 0; kc < nch; ++kc) {
            queries_begin();                              // stage g + 3 -> the slot stage g - 1 was read from (everybody is past that barrier)
            rows_begin();                                 // stage g + 2 -> likewise
            const uint4 *ab = lds + slot * SP3_A_UNITS + a_rd;
            const uint4 *bb = b_lds + bslot * SP_B_UNITS + b_rd;
            half8 bh[4], bl[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                bh[nt] = *reinterpret_cast<const half8 *>(bb + nt * 128);
                bl[nt] = *reinterpret_cast<const half8 *>(bb + nt * 128 + 64);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const half8 ah = *reinterpret_cast<const half8 *>(ab + mt * 128);
                const half8 al = *reinterpret_cast<const half8 *>(ab + mt * 128 + 64);
                // pair mode: x y = h h' + h l' + l h' (small terms first); half mode: the two planes are the two halves of a 64-float chunk
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, HALF ? bl[nt] : bh[nt], acc[mt][nt], 0, 0, 0);
                // the stage's six copy requests, in the order the wait below relies on (queries first), spread over the matrix work
                if (mt == 0) queries_piece(0);
                if (mt == 1) queries_piece(1);
                if (mt == 2) { rows_piece(0); rows_piece(1); }
                if (mt == 3) { rows_piece(2); rows_piece(3); }
                __builtin_amdgcn_sched_barrier(0);
                if (!HALF) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[nt], acc[mt][nt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[nt], acc[mt][nt], 0, 0, 0);
            }
            slot = slot + 1 == SP3_ARING ? 0 : slot + 1;
            bslot = (bslot + 1) & (SP3_BRING - 1);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // rows of stage g + 1 and queries of stage g + 2 have landed
            sp_stage_barrier();
        }
    }
    epilogue(tile_of(blockIdx.x + (my_tiles - 1) * gridDim.x));
    if (lane == 0) s.wcnt[blockIdx.x * (SP3_THREADS / 64) + (uint32_t)w] = wcount;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // nothing may land in LDS after the block is gone
}

// =====================================================================================================================================
// 256 queries per pass over the HALF copy.  The pass above is HBM-bound (6.05 TB/s of the ~6.3 the part delivers): the only way to more
// queries per second is fewer bytes per query, i.e. more queries per pass.  Here a stage is 128 rows (16 KiB from HBM: half of a 256-row
// tile of the copy, whose image is tile-of-16 major) against 256 queries (32 KiB from L2): the same 64 x 64 block of accumulators and the same
// 32 matrix instructions per wave and stage as above for half the HBM bytes.  8 waves = 2 (row halves) x 4 (query quarters); rings of three
// stages each (rows 48 KiB + queries 96 KiB = 144 KiB of LDS); per stage a wave requests [4 KiB of the queries, 2 KiB of the rows] of
// stage g + 2: six copies as above, so the same `s_waitcnt vmcnt(6)` means "stage g + 1 has landed".
// =====================================================================================================================================
#ifndef SP4_DBG
#define SP4_DBG 0          // QMX_TUNING builds: 1 = no copies at all (matrix loop alone), 3 = no query copies (wrong results)
#endif
constexpr int SP4_BM = 128;
constexpr int SP4_QT = 256;
constexpr int SP4_A_UNITS = SP4_BM * 2 * 4;                  // 1024 units = 16 KiB
constexpr int SP4_B_UNITS = SP4_QT * 2 * 4;                  // 2048 units = 32 KiB
constexpr int SP4_RING = 3;
constexpr int SP4_LDS = SP4_RING * (SP4_A_UNITS + SP4_B_UNITS) * 16;      // 144 KiB

__global__ __launch_bounds__(SP3_THREADS, 1) void scan_f16half256_kernel(const ScanArgs a, const SplitArgs s) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4 *lds = reinterpret_cast<uint4 *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t all_tiles = (a.n_cand + SP4_BM - 1) / SP4_BM;
    const uint64_t n_tiles = split_phase_tiles(all_tiles, s.phase);
    const uint32_t nch = s.nchunks;
    const uint64_t my_tiles = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const uint32_t phase = s.phase & 0xFFu, pstride = split_phase_stride(s.phase);
    auto tile_of = [&](uint64_t j) -> uint64_t { return phase == 0 ? j : phase == 1 ? j * pstride : j + j / (pstride - 1) + 1; };
    if (my_tiles == 0) {
        if (lane == 0) s.wcnt[blockIdx.x * (SP3_THREADS / 64) + (uint32_t)w] = 0;
        return;
    }
    const uint32_t wm = (uint32_t)w & 1u, wn = (uint32_t)w >> 1;
    const uint32_t kq_r = (uint32_t)lane >> 4, m_r = (uint32_t)lane & 15u;
    const uint32_t a_rd = sp_unit(wm * 4, 0, kq_r, m_r), b_rd = sp_unit(wn * 4, 0, kq_r, m_r);
    float thr[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) thr[nt] = s.thr[wn * 64 + nt * 16 + m_r];
    const float inv_scale = s.scales[2];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // from here on the kernel counts its vector-memory traffic itself
    const uint32_t lds0 = (uint32_t)(uintptr_t)(sp_lds_byte *)smem_raw;
    const uint32_t lane_off = (uint32_t)lane * 16u;
    uint4 *b_lds = lds + SP4_RING * SP4_A_UNITS;
    // The two copy streams advance by plain pointer increments (a stage of the rows is 32 KiB further in the copy, the tile's image being
    // contiguous over the K-chunks; the queries wrap after nch stages); the tile's address is computed once per tile.  The scalar unit
    // shares the issue port with the matrix instructions: ~90 scalar instructions per stage cost as much as the 32 MFMAs themselves.
    uint64_t ra_it = 0;
    uint32_t ra_kc = 0, ra_slot = 0, rb_kc = 0, rb_slot = 0;
    auto uniform_ptr = [&](uint64_t v) {
        return reinterpret_cast<const unsigned char *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
                                                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v));
    };
    auto tile_ptr = [&](uint64_t j) {      // first stage of the j-th tile of this block: half (tile & 1) of the copy's 256-row tile tile / 2
        const uint64_t tile = tile_of(blockIdx.x + j * gridDim.x);
        return uniform_ptr((uint64_t)(uintptr_t)(s.rows_split + (tile >> 1) * nch * SP3_A_UNITS + (tile & 1) * SP4_A_UNITS) + (uint32_t)w * 2048u);
    };
    const unsigned char *ra_cur = tile_ptr(0);
    const unsigned char *const rb_first = uniform_ptr((uint64_t)(uintptr_t)s.bq + (uint32_t)w * 4096u);
    const unsigned char *rb_cur = rb_first;
    const unsigned char *ra_src = nullptr, *rb_src = nullptr;
    uint32_t ra_dst = 0, rb_dst = 0;
    const uint32_t ra_dst0 = lds0 + (uint32_t)w * 2048u, rb_dst0 = lds0 + SP4_RING * SP4_A_UNITS * 16u + (uint32_t)w * 4096u;
    auto rows_begin = [&]() {
        ra_src = ra_cur;
        ra_dst = ra_dst0 + ra_slot * (SP4_A_UNITS * 16u);
        ra_slot = ra_slot + 1 == SP4_RING ? 0 : ra_slot + 1;
        if (ra_kc + 1 < nch) { ++ra_kc; ra_cur += SP3_A_UNITS * 16; }
        else if (ra_it + 1 < my_tiles) { ra_kc = 0; ++ra_it; ra_cur = tile_ptr(ra_it); }
    };
    auto rows_piece = [&](int i) {
#if SP4_DBG != 1
        sp_glds16_m0_nt(ra_src + i * 1024, lane_off, ra_dst + i * 1024);
#endif
    };
    auto queries_begin = [&]() {
        rb_src = rb_cur;
        rb_dst = rb_dst0 + rb_slot * (SP4_B_UNITS * 16u);
        rb_slot = rb_slot + 1 == SP4_RING ? 0 : rb_slot + 1;
        if (rb_kc + 1 == nch) { rb_kc = 0; rb_cur = rb_first; }
        else { ++rb_kc; rb_cur += SP4_B_UNITS * 16; }
    };
    auto queries_piece = [&](int i) {
#if SP4_DBG != 1 && SP4_DBG != 3
        sp_glds16_m0(rb_src + i * 1024, lane_off, rb_dst + i * 1024);
#endif
    };
    auto request_stage = [&]() {
        queries_begin();
#pragma unroll
        for (int i = 0; i < 4; ++i) queries_piece(i);
        rows_begin();
        rows_piece(0);
        rows_piece(1);
    };
    request_stage();                                      // stage 0
    request_stage();                                      // stage 1
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // stage 0 has landed
    sp_stage_barrier();
    uint32_t slot = 0;
    f32x4s acc[4][4];
    uint4 *const wl = s.wlist + (uint64_t)(blockIdx.x * (SP3_THREADS / 64) + (uint32_t)w) * s.wcap;
    uint32_t wcount = 0;
    auto epilogue = [&](uint64_t tile) {
        const uint32_t row0 = (uint32_t)(tile * SP4_BM) + wm * 64 + 4 * kq_r;
        const uint32_t n_rows32 = (uint32_t)a.n_cand;
        bool maybe = false;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            float mx = -__builtin_inff();
            bool nan = false;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    mx = __builtin_fmaxf(mx, acc[mt][nt][j]);
                    nan = nan || acc[mt][nt][j] != acc[mt][nt][j];
                }
            maybe = maybe || !(mx < thr[nt]) || nan;
        }
        if (!__ballot(maybe)) return;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const uint32_t q = wn * 64 + (uint32_t)nt * 16 + m_r;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = acc[mt][nt][j];
                    const uint32_t row = row0 + (uint32_t)mt * 16 + (uint32_t)j;
                    const bool c = !(v < thr[nt]) && row < n_rows32 && q < s.nq;
                    const uint64_t hits = __ballot(c);
                    if (hits) {
                        const uint32_t at = wcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(hits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hits, 0u));
                        if (c && at < s.wcap) {
                            const uint64_t key = make_key(v * inv_scale, row);
                            wl[at] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), q, 0u);
                        }
                        wcount += (uint32_t)__builtin_popcountll(hits);
                    }
                }
            }
    };
    // Operands are read ONE STAGE AHEAD: a stage's barrier sits after its third row tile (when the copies of stage g + 1 are known to have landed:
    // everything but the five requests this stage has issued by then), and the fourth tile is multiplied while the operands of stage g + 1
    // are already being read - the burst of 8 waves x 12 ds_read_b128 after a barrier no longer stands in front of idle matrix pipes.
    // Slot (g + 2) % 3 is free for the requests of stage g: its last readers finished before the barrier of stage g - 1.
    half8 bh[2][4], bl[2][4], ah[2], al[2];
    auto read_b = [&](auto Pc, uint32_t from_slot) {
        constexpr int P = decltype(Pc)::value;
        const uint4 *bb = b_lds + from_slot * SP4_B_UNITS + b_rd;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bl[P][nt] = *reinterpret_cast<const half8 *>(bb + nt * 128 + 64);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bh[P][nt] = *reinterpret_cast<const half8 *>(bb + nt * 128);
    };
    auto read_a = [&](int set, uint32_t from_slot, int mt) {
        const uint4 *ab = lds + from_slot * SP4_A_UNITS + a_rd + mt * 128;
        al[set] = *reinterpret_cast<const half8 *>(ab + 64);
        ah[set] = *reinterpret_cast<const half8 *>(ab);
    };
    read_a(0, 0, 0);
    read_b(std::integral_constant<int, 0>{}, 0);
    auto stage = [&](auto Pc) {
        constexpr int P = decltype(Pc)::value;
        queries_begin();                                  // stage g + 2 -> the slots stage g - 1 was read from
        rows_begin();
        const uint32_t next_slot = slot + 1 == SP4_RING ? 0 : slot + 1;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int cur = mt & 1, nxt = cur ^ 1;
            if (mt < 3) read_a(nxt, slot, mt + 1);
            else {
                read_a(nxt, next_slot, 0);                // (mt = 3 uses set 1: set 0 is free for the next stage's first tile)
                read_b(std::integral_constant<int, P ^ 1>{}, next_slot);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[cur], bl[P][nt], acc[mt][nt], 0, 0, 0);
            if (mt == 0) { queries_piece(0); queries_piece(1); }
            if (mt == 1) { queries_piece(2); queries_piece(3); }
            if (mt == 2) rows_piece(0);
            if (mt == 3) rows_piece(1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur], bh[P][nt], acc[mt][nt], 0, 0, 0);
            if (mt == 2) {
                asm volatile("s_waitcnt vmcnt(5)" ::: "memory");     // stage g + 1 has landed (this stage has requested five copies so far)
                sp_stage_barrier();
            }
        }
        slot = next_slot;
    };
    for (uint64_t it = 0; it < my_tiles; ++it) {
        if (it) epilogue(tile_of(blockIdx.x + (it - 1) * gridDim.x));
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4s){0.f, 0.f, 0.f, 0.f};
        for (uint32_t kc = 0; kc < nch; kc += 2) {        // (dim is a multiple of 128: an even number of 64-float stages)
            stage(std::integral_constant<int, 0>{});
            stage(std::integral_constant<int, 1>{});
        }
    }
    epilogue(tile_of(blockIdx.x + (my_tiles - 1) * gridDim.x));
    if (lane == 0) s.wcnt[blockIdx.x * (SP3_THREADS / 64) + (uint32_t)w] = wcount;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // nothing may land in LDS after the block is gone
}

// per-wave candidate lists -> per-query lists (what sp_select_kernel reads), deleted rows dropped.  One block per RG_LISTS wave lists: count per
// query in LDS, reserve the block's range of every query's list with ONE global atomic per query, then place.
constexpr int RG_LISTS = 16;
__global__ __launch_bounds__(256) void sp_regroup_kernel(const uint4 *wlist, const uint32_t *wcnt, uint32_t wcap, uint32_t n_lists, DeletedView del,
                                                         uint64_t *cand, uint32_t *cand_cnt, uint32_t cap, int *overflow /* of the tile: every query of it */) {
    __shared__ uint32_t hist[SP_QT_MAX], base[SP_QT_MAX];
    __shared__ uint32_t pre[RG_LISTS + 1];        // entries of the block's lists in front of list j
    if (threadIdx.x < SP_QT_MAX) hist[threadIdx.x] = 0;
    const uint32_t l0 = blockIdx.x * RG_LISTS, l1 = l0 + RG_LISTS < n_lists ? l0 + RG_LISTS : n_lists;
    if (threadIdx.x < 64) {                       // the lists' lengths in one load, their prefix sums across the wave
        const uint32_t l = l0 + threadIdx.x;
        uint32_t cnt = threadIdx.x < RG_LISTS && l < l1 ? wcnt[l] : 0;
        if (cnt > wcap) {
            *overflow = 1;
            cnt = wcap;
        }
        uint32_t inc = cnt;
        for (int o = 1; o < RG_LISTS; o <<= 1) {
            const uint32_t up = __shfl_up(inc, o, 64);
            if ((int)threadIdx.x >= o) inc += up;
        }
        if (threadIdx.x < RG_LISTS) pre[threadIdx.x + 1] = inc;
        if (threadIdx.x == 0) pre[0] = 0;
    }
    __syncthreads();
    const uint32_t total = pre[RG_LISTS];
    // the block's entries as one sequence: every trip of the loops below has 256 independent loads in flight (the lists are short - a few
    // dozen entries each - and walking them one after the other was a chain of dependent round trips: 25 us per launch, twice per step)
    auto entry = [&](uint32_t i) -> uint4 {
        uint32_t j = 0;
#pragma unroll
        for (int s2 = RG_LISTS / 2; s2 >= 1; s2 >>= 1)
            if (pre[j + s2] <= i) j += s2;
        return wlist[(uint64_t)(l0 + j) * wcap + (i - pre[j])];
    };
    for (uint32_t i = threadIdx.x; i < total; i += 256) {
        const uint4 e = entry(i);
        if (e.z < (uint32_t)SP_QT_MAX && del.live(e.x ^ 0xFFFFFFFFu)) atomicAdd(&hist[e.z], 1u);      // (e.z = 0xFFFFFFFF: an entry tq4w_finish_kernel withdrew)
    }
    __syncthreads();
    if (threadIdx.x < SP_QT_MAX) {
        const uint32_t c = hist[threadIdx.x];
        base[threadIdx.x] = c ? atomicAdd(&cand_cnt[threadIdx.x], c) : 0;
        hist[threadIdx.x] = 0;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < total; i += 256) {
        const uint4 e = entry(i);
        if (e.z >= (uint32_t)SP_QT_MAX || !del.live(e.x ^ 0xFFFFFFFFu)) continue;
        const uint32_t at = base[e.z] + atomicAdd(&hist[e.z], 1u);
        if (at < cap) cand[(uint64_t)e.z * cap + at] = ((uint64_t)e.y << 32) | e.x;
    }
}

// the pre-split copy: one thread per 16-byte unit of the OUTPUT, out[(tile * nch + kc) * 2048 + unit(row, h / l, kq)], in the output's order (whole
// lines written, as sp_i8_copy_kernel); rows past n are zero
__global__ __launch_bounds__(256) void sp_split_copy_kernel(const unsigned char *rows, uint64_t row_stride, uint64_t n, uint32_t dim, float scale, uint4 *out,
                                                            int half) {
    const uint32_t nch = half ? dim / 64 : dim / 32;
    const uint64_t n_tiles = (n + SP3_BM - 1) / SP3_BM;
    const uint64_t total = n_tiles * nch * SP3_A_UNITS;
    for (uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (uint64_t)gridDim.x * 256) {
        const uint32_t u = (uint32_t)(gid % SP3_A_UNITS);
        const uint64_t blk = gid / SP3_A_UNITS, tile = blk / nch;
        const uint32_t kc = (uint32_t)(blk % nch);
        // sp_unit(t, hl, kq, m) = ((t * 2 + hl) * 4 + kq) * 16 + (m ^ 2 kq), inverted
        const uint32_t kq = (u >> 4) & 3u, hl = (u >> 6) & 1u, t = u >> 7, m = (u & 15u) ^ (2u * kq);
        const uint64_t r = tile * SP3_BM + t * 16u + m;
        // half: a plane pair per 64 floats = the high parts of its two 32-float halves; else the high (hl = 0) and low (hl = 1) parts of 32 floats
        const uint32_t k32 = half ? kc * 2u + hl : kc;
        half8 h, l;
        if (r < n) {
            const float *v = reinterpret_cast<const float *>(rows + r * row_stride) + k32 * 32 + kq * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                h[e] = (_Float16)__builtin_fmaf(v[e], scale, 0.0f);
                l[e] = (_Float16)__builtin_fmaf(v[e], scale, -(float)h[e]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { h[e] = (_Float16)0.0f; l[e] = (_Float16)0.0f; }
        }
        out[gid] = (half || hl == 0) ? *reinterpret_cast<const uint4 *>(&h) : *reinterpret_cast<const uint4 *>(&l);
    }
}

// ---- candidates -> the rows worth an exact score: A_k = k-th best approximate key, keep A >= A_k - 2 band (at most vcap per query) ----
constexpr int SEL_BLOCK = 512;
// k-th best of raw keys (0 when there are fewer than k), whole block; the result is valid in wave 0 (and broadcast through *sh_out after the sync)
__device__ __forceinline__ void block_kth_key(const uint64_t *c, uint32_t raw, int ptop, uint64_t (*sh)[WAVE], uint64_t *sh_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t list = 0;
    for (uint32_t base = wave * WAVE; base < raw; base += SEL_BLOCK) {
        const uint32_t i = base + lane;
        uint64_t key = i < raw ? c[i] : 0;
        if (key <= readlane_u64(list, ptop - 1)) key = 0;
        uint64_t m = __ballot(key != 0);
        while (m) {
            const int src = __builtin_ctzll(m);
            m &= m - 1;
            const uint64_t nk = readlane_u64(key, src);
            if (nk > readlane_u64(list, ptop - 1)) wave_list_insert(list, nk, lane);
        }
    }
    sh[wave][lane] = list;
    __syncthreads();
    if (wave == 0) {
        uint64_t merged = sh[0][lane];
        for (int w2 = 1; w2 < SEL_BLOCK / WAVE; ++w2) {
            const uint64_t key = sh[w2][lane];
            uint64_t m = __ballot(key > readlane_u64(merged, ptop - 1));
            while (m) {
                const int src = __builtin_ctzll(m);
                m &= m - 1;
                const uint64_t nk = readlane_u64(key, src);
                if (nk > readlane_u64(merged, ptop - 1)) wave_list_insert(merged, nk, lane);
            }
        }
        if (lane == 0) *sh_out = readlane_u64(merged, ptop - 1);
    }
    __syncthreads();
}

// The same for lists of up to 8 keys per thread (the usual ~1 000 candidates of a query) without serial insertions, as custom_topk_small_kernel does it:
// the k-th largest of a wave's 64 lane maxima bounds the k-th best key from below, the largest such bound over the waves prunes the list to a few
// times k survivors in LDS, and the survivor with k - 1 larger ones is the answer (keys are distinct: they carry the row id).
constexpr int SEL_E = 3;       // (8 until round 5: 32 KiB of scratch; 3: 12 KiB - a selection block fits the 16 KiB the int8 scan leaves free on a CU)
constexpr int SEL_SURV = SEL_BLOCK * SEL_E;
struct SelScratch {
    uint64_t surv[SEL_SURV];
    uint64_t t[SEL_BLOCK / WAVE];
    uint32_t cnt;
};
// the serial-insertion path (block_kth_key) and the ranked one (block_kth_key_ranked) never run in the same block: one LDS area for both
union SelShared {
    uint64_t sh[SEL_BLOCK / WAVE][WAVE];
    SelScratch scratch;
};
__device__ __forceinline__ void block_kth_key_ranked(const uint64_t *c, uint32_t raw, uint32_t ptop, SelScratch *sc, uint64_t *sh_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t kreg[SEL_E];
    uint64_t m = 0;
#pragma unroll
    for (int e = 0; e < SEL_E; ++e) {
        const uint32_t i = (uint32_t)e * SEL_BLOCK + threadIdx.x;
        kreg[e] = i < raw ? c[i] : 0ull;
        m = kreg[e] > m ? kreg[e] : m;
    }
    uint32_t rank = 0;
    for (int j = 0; j < WAVE; ++j) rank += readlane_u64(m, j) > m ? 1u : 0u;
    const uint64_t sel = __ballot(rank == ptop - 1 && m != 0);
    const uint64_t tw = sel ? readlane_u64(m, __builtin_ctzll(sel)) : 0ull;
    if (lane == 0) sc->t[wave] = tw;
    if (threadIdx.x == 0) { sc->cnt = 0; *sh_out = 0; }
    __syncthreads();
    uint64_t t = 0;
#pragma unroll
    for (int w = 0; w < SEL_BLOCK / WAVE; ++w) t = sc->t[w] > t ? sc->t[w] : t;
#pragma unroll
    for (int e = 0; e < SEL_E; ++e)
        if (kreg[e] != 0 && kreg[e] >= t) sc->surv[atomicAdd(&sc->cnt, 1u)] = kreg[e];
    __syncthreads();
    const uint32_t cnt = sc->cnt;
    for (uint32_t i = threadIdx.x; i < cnt; i += SEL_BLOCK) {
        const uint64_t my = sc->surv[i];
        uint32_t r = 0;
        for (uint32_t j = 0; j < cnt; ++j) r += (sc->surv[j] > my || (sc->surv[j] == my && j < i)) ? 1u : 0u;
        if (r == ptop - 1) *sh_out = my;
    }
    __syncthreads();
}

// After the strided sixteenth of the block: the k-th best APPROXIMATE score A among its rows proves k rows with an exact score >= A - band,
// so a result row has an approximate score >= A - 2 band: the threshold of the other fifteen sixteenths (never lowered).
__global__ __launch_bounds__(SEL_BLOCK) void sp_refine_kernel(const uint64_t *cand, const uint32_t *cand_cnt, uint32_t cap, const float *band, uint32_t top,
                                                              const float *scales, float *thr) {
    __shared__ SelShared sel_sh;
    uint64_t (*const sh)[WAVE] = sel_sh.sh;
    SelScratch &scratch = sel_sh.scratch;
    __shared__ uint64_t sh_kth;
    const uint32_t q = blockIdx.x;
    const uint32_t raw = cand_cnt[q];
    if (raw > cap || raw < top || !(band[q] < 3.0e38f)) return;   // overflow is sp_select_kernel's to report; fewer than k rows prove nothing
    if (raw <= (uint32_t)SEL_SURV) block_kth_key_ranked(cand + (uint64_t)q * cap, raw, top, &scratch, &sh_kth);
    else block_kth_key(cand + (uint64_t)q * cap, raw, (int)top, sh, &sh_kth);
    if (threadIdx.x == 0 && sh_kth) {
        const float t = (key_score(sh_kth) - 2.0f * band[q]) * scales[1];
        if (t > thr[q]) thr[q] = t;
    }
}

__global__ __launch_bounds__(SEL_BLOCK) void sp_select_kernel(const uint64_t *cand, const uint32_t *cand_cnt, uint32_t cap, const float *band, uint32_t top,
                                                              const VerifyPool pool, uint32_t q_base, const int *tile_overflow, uint32_t *ovf_q,
                                                              SplitStats *stats, const float *t_exact /* or nullptr: an exact lower bound of the k-th best score per query */,
                                                              bool tighten /* with t_exact: the k-th best approximate score's bound as well */) {
    __shared__ SelShared sel_sh;
    uint64_t (*const sh)[WAVE] = sel_sh.sh;
    SelScratch &scratch = sel_sh.scratch;
    __shared__ uint64_t sh_kth;
    __shared__ float sh_cut;
    __shared__ uint32_t sh_n, sh_off;
    const uint32_t q = blockIdx.x;
    const uint32_t raw = cand_cnt[q];
    // a wave list of the scan overflowed (its dropped entries could be anybody's), or this query's candidate buffer did: the pass cannot be
    // trusted for this query, which takes the exact scan (the other queries of the batch keep their lists)
    if (*tile_overflow || raw > cap || !(band[q] < 3.0e38f)) {      // (an infinite band: the query had no usable threshold)
        if (threadIdx.x == 0) { ovf_q[q] = 1; pool.cnt[q_base + q] = 0; pool.off[q_base + q] = 0; }
        return;
    }
    const uint64_t *c = cand + (uint64_t)q * cap;
    if (threadIdx.x == 0) sh_n = 0;
    // with an exact bound T (the int8 copy's passes): a result row's approximate score is >= T - band; without: the k-th best approximate score A_k
    // proves k rows with an exact score >= A_k - band, so >= A_k - 2 band
    const bool have_t = t_exact != nullptr && t_exact[q] > -__builtin_inff();
    if (have_t && tighten) {
        // both bounds (the 128-query TurboQuant pass: T comes from a sample alone, so A_k - 2 band is usually the tighter cut)
        if (raw <= (uint32_t)SEL_SURV) block_kth_key_ranked(c, raw, top, &scratch, &sh_kth);
        else block_kth_key(c, raw, (int)top, sh, &sh_kth);
        if (threadIdx.x == 0) {
            const float by_t = t_exact[q] - band[q], by_k = sh_kth ? key_score(sh_kth) - 2.0f * band[q] : -__builtin_inff();
            sh_cut = by_k > by_t ? by_k : by_t;
        }
    } else if (have_t) {
        if (threadIdx.x == 0) sh_cut = t_exact[q] - band[q];
    } else {
        if (raw <= (uint32_t)SEL_SURV) block_kth_key_ranked(c, raw, top, &scratch, &sh_kth);
        else block_kth_key(c, raw, (int)top, sh, &sh_kth);
        // fewer than k candidates: keep all of them (cut = -inf)
        if (threadIdx.x == 0) sh_cut = sh_kth ? key_score(sh_kth) - 2.0f * band[q] : -__builtin_inff();
    }
    __syncthreads();
    const float cut = sh_cut;
    // how many rows the query verifies, then its range of the pool (one global atomic per query), then the rows
    uint32_t mine = 0;
    for (uint32_t i = threadIdx.x; i < raw; i += SEL_BLOCK) mine += !(key_score(c[i]) < cut) ? 1u : 0u;
    if (mine) atomicAdd(&sh_n, mine);
    __syncthreads();
    const uint32_t n = sh_n;
    if (threadIdx.x == 0) {
        const bool too_many = n > pool.max_per_query;                     // (more rows than a query may verify: it reserves nothing)
        const uint32_t off = (n && !too_many) ? atomicAdd(pool.used, n) : 0u;
        const bool over = too_many || (n != 0 && ((uint64_t)off + n > pool.cap));     // (or the pool is full: this query - alone - takes the exact scan)
        sh_off = over ? 0xFFFFFFFFu : off;
        sh_kth = too_many ? (uint64_t)pool.cap : (uint64_t)off;           // (the raw offset: what a query that ran over the pool's end must fill, below)
        ovf_q[q] = over ? 1u : 0u;
        pool.off[q_base + q] = over ? 0u : off;
        pool.cnt[q_base + q] = over ? 0u : n;
        atomicAdd(&stats->candidates, (unsigned long long)raw);
        if (!over) atomicAdd(&stats->verified, (unsigned long long)n);
        sh_n = 0;
    }
    __syncthreads();
    const uint32_t off = sh_off;
    if (off == 0xFFFFFFFFu) {
        // what this query reserved inside the pool before it ran over stays in front of `used`: the gather visits those entries, so they name a row
        // that exists (row 0: its score is dropped - the query's count is zero)
        const uint64_t lo = sh_kth;
        for (uint64_t i = lo + threadIdx.x; i < pool.cap; i += SEL_BLOCK) {
            pool.ids[i] = 0;
            pool.qsel[i] = q_base + q;
        }
        return;
    }
    if (n == 0) return;
    for (uint32_t i = threadIdx.x; i < raw; i += SEL_BLOCK) {
        const uint64_t key = c[i];
        if (!(key_score(key) < cut)) {
            const uint32_t slot = off + atomicAdd(&sh_n, 1u);
            pool.ids[slot] = key_idx(key);
            pool.qsel[slot] = q_base + q;
        }
    }
}

// ---- the queries whose lists overflowed, packed: list[j] = j-th such query (0xFFFFFFFF behind the last), its pre-scan bound next to it; the run flags of
// the conditional exact passes (api_*.hip: one 16-query pass when 1..16 queries overflowed, else passes of 64) ----
__global__ __launch_bounds__(256) void sp_plan_kernel(const uint32_t *ovf_q, uint32_t nq, const uint64_t *gthr, uint32_t *list, uint64_t *gthr_packed,
                                                       uint32_t list_cap, uint32_t *count, int *run16, int *run64, uint32_t n_run64, SplitStats *stats,
                                                       const uint4 *queries, uint32_t units_per_query, uint4 *packed_queries) {
    __shared__ uint32_t sh_cnt;
    const uint32_t lane = threadIdx.x & 63;
    if (threadIdx.x < 64) {
        uint32_t cnt = 0;
        for (uint32_t base = 0; base < nq; base += 64) {
            const uint32_t i = base + lane;
            const bool f = i < nq && ovf_q[i] != 0;
            const uint64_t m = __ballot(f);
            if (f) {
                const uint32_t at = cnt + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                list[at] = i;
                gthr_packed[at] = gthr[i];
            }
            cnt += (uint32_t)__popcll(m);
        }
        for (uint32_t i = cnt + lane; i < list_cap; i += 64) { list[i] = 0xFFFFFFFFu; gthr_packed[i] = 0; }
        if (lane == 0) {
            *count = cnt;
            *run16 = cnt >= 1 && cnt <= 16;
            const uint32_t fqt = n_run64 ? list_cap / n_run64 : 64u;      // queries per conditional pass: 64, or 32 for rows the 64-query shape does not take
            for (uint32_t p = 0; p < n_run64; ++p) run64[p] = cnt > (p ? fqt * p : 16u);
            stats->fallback_queries = cnt;
            sh_cnt = cnt;
        }
    }
    __syncthreads();
    const uint32_t cnt = sh_cnt;
    if (cnt == 0) return;
    // their query entries, packed the same way (slots behind the last repeat the first: valid operands, results dropped); a rare path, one block does it
    __threadfence_block();
    for (uint64_t gid = threadIdx.x; gid < (uint64_t)list_cap * units_per_query; gid += 256) {
        const uint32_t j = (uint32_t)(gid / units_per_query), u = (uint32_t)(gid % units_per_query);
        const uint32_t src = list[j < cnt ? j : 0];
        packed_queries[gid] = queries[(uint64_t)src * units_per_query + u];
    }
}

// ---- row statistics of an f32 block (once per segment): stats[0] = max |x|, stats[1] = max row sum of squares (uint bits) ----
__global__ __launch_bounds__(256) void sp_row_stats_kernel(const unsigned char *rows, uint64_t row_stride, uint64_t n, uint32_t dim, uint32_t *stats) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (uint64_t)gridDim.x * 4;
    float mx = 0.0f, mss = 0.0f;
    for (uint64_t r = wave; r < n; r += nwaves) {
        const float *v = reinterpret_cast<const float *>(rows + r * row_stride);
        float ss = 0.0f;
        for (uint32_t i = lane; i < dim; i += 64) {
            const float x = v[i];
            mx = __builtin_fmaxf(mx, __builtin_fabsf(x));
            ss = __builtin_fmaf(x, x, ss);
        }
        for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
        mss = __builtin_fmaxf(mss, ss);
    }
    for (int o = 32; o >= 1; o >>= 1) mx = __builtin_fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) {
        atomicMax(&stats[0], __float_as_uint(mx));
        atomicMax(&stats[1], __float_as_uint(mss));
    }
}

// =====================================================================================================================================
// The INT8 copy (QMX_SEG_I8_COPY): one byte per element, the same LDS images, the same staging - half the bytes of the half copy per query.
//   rows     x_i = s_i (c_i + e_i),  s_i = max_r |x_ri| / 127 (one scale per COLUMN: an outlier coordinate costs the others nothing, and the
//            scale folds into the query), c_i = rint(x_i / s_i) in [-127, 127], |e_i| <= 1/2
//   queries  q_i s_i = t (d_i + f_i),  t = max_i |q_i s_i| / 127 (one scale per QUERY), d_i = rint(q_i s_i / t), |f_i| <= 1/2
//   score    sum x_i q_i = t [ sum c_i d_i  +  sum c_i f_i + sum e_i d_i + sum e_i f_i ]
// The first sum is the integer the matrix cores deliver (v_mfma_i32_16x16x64_i8, exact); the other three are bounded, worst case, by
//   band_q = t [ min(C1 / 2, C2 |f|_2) + (sum |d_i|) / 2 + dim / 4 ],   C1 = max_r sum_i |c_ri|,  C2 = max_r |c_r|_2   (taken once per segment)
// (+ the round-off of the f32 evaluation the exact scores carry).  On unit Gaussian rows that is ~0.7 standard deviations of the score - two
// orders above the error that really occurs, the price of a bound nobody can break - so the pass works with EXACT lower bounds T_q of the
// k-th best score instead of approximate ones: a row of the result has an approximate score >= T_q - band_q (one band, not two).  T_q comes from
// the exact sample scores first, then - after each of the two launches - from the exact scores of the k best candidates so far
// (sp_i8_probe_kernel, the gather kernel of qmx_rescore, sp_i8_bound_kernel): ~a few hundred rows per query are re-scored at the end.
// Everything else (per-wave candidate lists, regroup, verification, per-query exact fallback) is the f16 path's.
// =====================================================================================================================================
typedef int i32x4s __attribute__((ext_vector_type(4)));
constexpr uint32_t SP_I8_PROBE = 64;                         // slots per query for the candidates whose exact scores renew the bound after a launch (k <= 64 of them)
constexpr uint32_t SP_I8_MAX_DIM = 4096;                     // (sp_i8_colmax_kernel keeps a column maximum per thread and 256 columns)

// column maxima and sums of squares: colmax[c] = max_r |x_rc| (uint bits of a non-negative float order like the float), colsq[c] = sum_r x_rc^2 (f32, a
// statistic that only steers the choice of the scales); thread t owns columns t, t + 256, ...
__global__ __launch_bounds__(256) void sp_i8_colmax_kernel(const unsigned char *rows, uint64_t row_stride, uint64_t n, uint32_t dim, uint32_t *colmax, float *colsq) {
    float mx[SP_I8_MAX_DIM / 256], sq[SP_I8_MAX_DIM / 256];
#pragma unroll
    for (int j = 0; j < (int)(SP_I8_MAX_DIM / 256); ++j) { mx[j] = 0.0f; sq[j] = 0.0f; }
    for (uint64_t r = blockIdx.x; r < n; r += gridDim.x) {
        const float *v = reinterpret_cast<const float *>(rows + r * row_stride);
#pragma unroll
        for (int j = 0; j < (int)(SP_I8_MAX_DIM / 256); ++j) {
            const uint32_t c = threadIdx.x + 256u * (uint32_t)j;
            if (c < dim) {
                const float x = v[c];
                mx[j] = __builtin_fmaxf(mx[j], __builtin_fabsf(x));
                sq[j] = __builtin_fmaf(x, x, sq[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < (int)(SP_I8_MAX_DIM / 256); ++j) {
        const uint32_t c = threadIdx.x + 256u * (uint32_t)j;
        if (c < dim) {
            atomicMax(&colmax[c], __float_as_uint(mx[j]));
            atomicAdd(&colsq[c], sq[j]);
        }
    }
}
__device__ __forceinline__ int sp_i8_code(float x, float s) {
    int c = (int)__builtin_rintf(x / s);
    return c > 127 ? 127 : (c < -127 ? -127 : c);
}
// stats[0] = C1 = max_r sum |c_ri|, stats[1] = C2^2 = max_r sum c_ri^2, stats[2] != 0: an element that is not finite (no int8 copy for this block)
__global__ __launch_bounds__(256) void sp_i8_row_stats_kernel(const unsigned char *rows, uint64_t row_stride, uint64_t n, uint32_t dim, const float *scale,
                                                              uint32_t *stats) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (uint64_t)gridDim.x * 4;
    uint32_t m1 = 0, m2 = 0;
    bool bad = false;
    for (uint64_t r = wave; r < n; r += nwaves) {
        const float *v = reinterpret_cast<const float *>(rows + r * row_stride);
        uint32_t c1 = 0, c2 = 0;
        for (uint32_t i = lane; i < dim; i += 64) {
            const float x = v[i];
            bad = bad || !(__builtin_fabsf(x) < __builtin_inff());
            const int c = sp_i8_code(x, scale[i]);
            c1 += (uint32_t)(c < 0 ? -c : c);
            c2 += (uint32_t)(c * c);
        }
        for (int o = 32; o >= 1; o >>= 1) {
            c1 += __shfl_xor(c1, o, 64);
            c2 += __shfl_xor(c2, o, 64);
        }
        m1 = c1 > m1 ? c1 : m1;
        m2 = c2 > m2 ? c2 : m2;
    }
    if (lane == 0) {
        atomicMax(&stats[0], m1);
        atomicMax(&stats[1], m2);
    }
    if (bad) atomicOr(&stats[2], 1u);
}
// the copy: one thread per 16-coordinate unit of the OUTPUT, out[(tile * nch + k128) * 2048 + unit(row, plane = which 64 of the 128, kq)], in the output's
// order - a wave writes 1 KiB of consecutive units (whole lines: the row-major order of round 3 wrote 13.3 GB for a 7.68 GB copy, profiles/
// r3_pmc_traffic_i8.md) and reads sixteen rows' 256-byte runs; rows past n are zero
__global__ __launch_bounds__(256) void sp_i8_copy_kernel(const unsigned char *rows, uint64_t row_stride, uint64_t n, uint32_t dim, const float *scale, uint4 *out) {
    const uint32_t nch = dim / 128;
    const uint64_t n_tiles = (n + SP3_BM - 1) / SP3_BM;
    const uint64_t total = n_tiles * nch * SP3_A_UNITS;
    for (uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (uint64_t)gridDim.x * 256) {
        const uint32_t u = (uint32_t)(gid % SP3_A_UNITS);
        const uint64_t blk = gid / SP3_A_UNITS, tile = blk / nch;
        const uint32_t k128 = (uint32_t)(blk % nch);
        // sp_unit(t, hl, kq, m) = ((t * 2 + hl) * 4 + kq) * 16 + (m ^ 2 kq), inverted
        const uint32_t kq = (u >> 4) & 3u, hl = (u >> 6) & 1u, t = u >> 7, m = (u & 15u) ^ (2u * kq);
        const uint64_t r = tile * SP3_BM + t * 16u + m;
        const uint32_t g = k128 * 8u + hl * 4u + kq;
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        if (r < n) {
            const float *v = reinterpret_cast<const float *>(rows + r * row_stride) + g * 16;
            const float *sc = scale + g * 16;
#pragma unroll
            for (int e = 0; e < 16; ++e) w[e / 4] |= ((uint32_t)sp_i8_code(v[e], sc[e]) & 0xFFu) << (8 * (e % 4));
        }
        out[gid] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ---- once per 128-query tile: one block per query slot.  The query in the columns' scales, its own scale, its codes in the B-operand images
// (bq[k128][unit(query, plane, kq)], zeros past nq), its band, its first threshold (from the sample's exact k-th best score) ----
__device__ __forceinline__ float sp_block_max(float v, float *sh) {
    for (int o = 32; o >= 1; o >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return __builtin_fmaxf(__builtin_fmaxf(sh[0], sh[1]), __builtin_fmaxf(sh[2], sh[3]));
}
__device__ __forceinline__ float sp_block_sum(float v, float *sh) {
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ __launch_bounds__(256) void sp_i8_pack_kernel(const float *q, uint32_t nq, uint32_t dim, const float *scale, const uint64_t *gthr,
                                                         const uint32_t *row_stats, float row_norm_max, uint4 *bq, float *qscale, float *band, float *thr,
                                                         float *t_exact, uint32_t *cand_cnt, uint32_t n_cnt) {
    extern __shared__ __attribute__((aligned(16))) float sh_q[];      // the query in the columns' scales
    __shared__ float sh_red[4];
    const uint32_t qi = blockIdx.x;
    const bool live = qi < nq;
    if (qi == 0)
        for (uint32_t i = threadIdx.x; i < n_cnt; i += 256) cand_cnt[i] = 0;
    float mx = 0.0f, ss = 0.0f, bad = 0.0f;
    for (uint32_t i = threadIdx.x; i < dim; i += 256) {
        const float raw = live ? q[(uint64_t)qi * dim + i] : 0.0f;
        const float v = raw * scale[i];
        sh_q[i] = v;
        mx = __builtin_fmaxf(mx, __builtin_fabsf(v));
        ss = __builtin_fmaf(raw, raw, ss);
        if (!(__builtin_fabsf(v) < __builtin_inff())) bad = 1.0f;
    }
    mx = sp_block_max(mx, sh_red);
    ss = sp_block_sum(ss, sh_red);
    bad = sp_block_max(bad, sh_red);                                  // (the barriers inside also publish sh_q)
    const float t = mx > 1.0e-30f ? mx / 127.0f : 1.0f;
    float sabs = 0.0f, sf2 = 0.0f;                                    // sum |d_i| (an integer below 2^24: exact in f32), sum f_i^2
    for (uint32_t u = threadIdx.x; u < dim / 16; u += 256) {
        const uint32_t k128 = u / 8, hl = (u / 4) & 1u, kq = u % 4;
        uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float x = sh_q[u * 16 + e] / t;
            float c = __builtin_rintf(x);
            c = c > 127.0f ? 127.0f : (c < -127.0f ? -127.0f : c);
            if (!(c == c)) c = 0.0f;
            const float f = x - c;
            sabs += __builtin_fabsf(c);
            sf2 = __builtin_fmaf(f, f, sf2);
            w[e / 4] |= ((uint32_t)(int)c & 0xFFu) << (8 * (e % 4));
        }
        bq[(uint64_t)k128 * SP_B_UNITS + sp_unit(qi >> 4, hl, kq, qi & 15u)] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    sabs = sp_block_sum(sabs, sh_red);
    sf2 = sp_block_sum(sf2, sh_red);
    if (threadIdx.x != 0) return;
    if (!live) {
        qscale[qi] = 0.0f;
        band[qi] = 0.0f;
        thr[qi] = __builtin_inff();
        t_exact[qi] = -__builtin_inff();
        return;
    }
    const float c1 = (float)row_stats[0], c2 = __builtin_sqrtf((float)row_stats[1]);
    const float cf = __builtin_fminf(0.5f * c1, c2 * __builtin_sqrtf(sf2));
    // (1.001: the roundings of x / s, q s, q s / t and of this sum; the last term: the f32 evaluation of the exact score, dim 2^-23 |q| |row|, twice)
    const float b = t * (cf + 0.5f * sabs + 0.25f * (float)dim) * 1.001f + 2.0f * (float)dim * 1.1920929e-7f * row_norm_max * __builtin_sqrtf(ss);
    const uint64_t k = gthr[qi];
    const bool ok = bad == 0.0f && k != 0 && b < 3.0e38f;
    // no bound (the sample holds fewer than k live rows) or a query that is not finite: no candidates, and the infinite band sends the query - alone -
    // to the exact scan
    qscale[qi] = t;
    band[qi] = ok ? b : __builtin_inff();
    t_exact[qi] = ok ? key_score(k) : -__builtin_inff();
    thr[qi] = ok ? (key_score(k) - b) / t : __builtin_inff();
}

// The scan: scan_f16pair_kernel<true> with the int8 instruction.  A stage is 256 rows x 128 coordinates (two planes of 64) = 32 KiB of rows + 16 KiB of
// queries, 32 matrix instructions per wave as there; nch = dim / 128 stages per tile.  s.scales = the queries' scales (score units per accumulator
// unit), s.thr in accumulator units.
// SP8_BRING = 3 query stages in LDS: the queries of stage g + 2 are requested during stage g (the 96 KiB of query images stay in L2: 2.9 us ahead is plenty),
// 96 + 48 = 144 KiB - which leaves 16 KiB of every CU's LDS to the OTHER batches' small kernels (regroup, probe, exact gather, bound, select, sort, pack ...):
// they run beside this scan instead of queueing behind it.  Rounds 3 - 5 kept four stages (160 KiB, the CU's whole LDS: nothing that needs LDS could start
// on a CU while a block of the scan lived there); measured in round 6 (C2, 128 queries per step; profiles/r6_i8_lanes_experiment.txt): 2 / 3 / 4 batches in
// flight 87.5 / 88.1 / 88.6 k QPS with 160 KiB, 88.3 / 90.0 / 91.4 k with 144 KiB.
constexpr int SP8_BRING = 3;
constexpr int SP8_LDS = (SP3_ARING * SP3_A_UNITS + SP8_BRING * SP_B_UNITS) * 16;
__global__ __launch_bounds__(SP3_THREADS, 1) void scan_i8copy_kernel(const ScanArgs a, const SplitArgs s) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4 *lds = reinterpret_cast<uint4 *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t all_tiles = (a.n_cand + SP3_BM - 1) / SP3_BM;
    const uint64_t n_tiles = split_phase_tiles(all_tiles, s.phase);
    const uint32_t nch = s.nchunks;
    const uint64_t my_tiles = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const uint32_t phase = s.phase & 0xFFu, pstride = split_phase_stride(s.phase);
    auto tile_of = [&](uint64_t j) -> uint64_t { return phase == 0 ? j : phase == 1 ? j * pstride : j + j / (pstride - 1) + 1; };
    if (my_tiles == 0) {
        if (lane == 0) s.wcnt[blockIdx.x * (SP3_THREADS / 64) + (uint32_t)w] = 0;
        return;
    }
    const uint32_t wm = (uint32_t)w & 3u, wn = (uint32_t)w >> 2;
    const uint32_t kq_r = (uint32_t)lane >> 4, m_r = (uint32_t)lane & 15u;
    const uint32_t a_rd = sp_unit(wm * 4, 0, kq_r, m_r), b_rd = sp_unit(wn * 4, 0, kq_r, m_r);
    float thr[4], qs[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        thr[nt] = s.thr[wn * 64 + nt * 16 + m_r];
        qs[nt] = s.scales[wn * 64 + nt * 16 + m_r];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // from here on the kernel counts its vector-memory traffic itself
    const uint32_t lds0 = (uint32_t)(uintptr_t)(sp_lds_byte *)smem_raw;
    const uint32_t lane_off = (uint32_t)lane * 16u;
    uint4 *b_lds = lds + SP3_ARING * SP3_A_UNITS;
    // the copy streams and their bookkeeping: scan_f16pair_kernel's, word for word (stage g requests [queries of g + 3, rows of g + 2], six 1 KiB
    // copies per wave, `vmcnt(6)` = the rows of g + 1 and the queries of g + 2 have landed)
    uint64_t ra_it = 0;
    uint32_t ra_kc = 0, ra_slot = 0, rb_kc = 0, rb_slot = 0;
    auto uniform_ptr = [&](uint64_t v) {
        return reinterpret_cast<const unsigned char *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
                                                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v));
    };
    const unsigned char *ra_src = nullptr, *rb_src = nullptr;
    uint32_t ra_dst = 0, rb_dst = 0;
    auto rows_begin = [&]() {
        const uint64_t tile = tile_of(blockIdx.x + ra_it * gridDim.x);
        ra_src = uniform_ptr((uint64_t)(uintptr_t)(s.rows_split + (tile * nch + ra_kc) * SP3_A_UNITS) + (uint32_t)w * 4096u);
        ra_dst = lds0 + (ra_slot * SP3_A_UNITS) * 16u + (uint32_t)w * 4096u;
        ra_slot = ra_slot + 1 == SP3_ARING ? 0 : ra_slot + 1;
        if (ra_kc + 1 < nch) ++ra_kc;
        else if (ra_it + 1 < my_tiles) { ra_kc = 0; ++ra_it; }
    };
    // (the rows non-temporal - read once, by this CU: 0.645 - 0.652 ms per launch against 0.661 - 0.670 with the default policy; the queries' slices, which every
    // block re-reads from L2, keep it)
    auto rows_piece = [&](int i) { sp_glds16_nt(ra_src + i * 1024, lane_off, ra_dst + i * 1024); };
    auto queries_begin = [&]() {
        rb_src = uniform_ptr((uint64_t)(uintptr_t)(s.bq + (uint64_t)rb_kc * SP_B_UNITS) + (uint32_t)w * 2048u);
        rb_dst = lds0 + (SP3_ARING * SP3_A_UNITS + rb_slot * SP_B_UNITS) * 16u + (uint32_t)w * 2048u;
        rb_slot = rb_slot + 1 == SP8_BRING ? 0 : rb_slot + 1;
        rb_kc = rb_kc + 1 == nch ? 0 : rb_kc + 1;
    };
    auto queries_piece = [&](int i) { sp_glds16(rb_src + i * 1024, lane_off, rb_dst + i * 1024); };
    auto request_rows = [&]() {
        rows_begin();
#pragma unroll
        for (int i = 0; i < 4; ++i) rows_piece(i);
    };
    auto request_queries = [&]() {
        queries_begin();
        queries_piece(0);
        queries_piece(1);
    };
    request_queries();                                    // queries of stage 0   (stage g requests [queries of g + 2, rows of g + 2])
    request_rows();                                       // rows of stage 0
    request_queries();                                    // queries of stage 1
    request_rows();                                       // rows of stage 1
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // queries and rows of stage 0 have landed
    sp_stage_barrier();
    uint32_t slot = 0, bslot = 0;
    i32x4s acc[4][4];
    uint4 *const wl = s.wlist + (uint64_t)(blockIdx.x * (SP3_THREADS / 64) + (uint32_t)w) * s.wcap;
    uint32_t wcount = 0;
    // The epilogue of a tile.  With a band of 0.7 standard deviations a wave meets ~2 candidates per tile (the f16 passes: 0.4), so nearly every tile
    // has one somewhere: the search for them narrows by wave-uniform steps - the query tile (16 queries x the wave's 64 rows), then the 16-row tile, then
    // the four rows of a lane - instead of testing all 256 accumulators of a lane one ballot at a time (14 % of the kernel at 128 queries, measured
    // against the same launch with one live query).
    auto epilogue = [&](uint64_t tile) {
        const uint32_t row0 = (uint32_t)(tile * SP3_BM) + wm * 64 + 4 * kq_r;
        const uint32_t n_rows32 = (uint32_t)a.n_cand;
        bool hit[4];
        bool maybe = false;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            int mx = acc[0][nt][0];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) mx = acc[mt][nt][j] > mx ? acc[mt][nt][j] : mx;
            hit[nt] = !((float)mx < thr[nt]);
            maybe = maybe || hit[nt];
        }
        if (!__ballot(maybe)) return;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            if (!__ballot(hit[nt])) continue;
            const uint32_t q = wn * 64 + (uint32_t)nt * 16 + m_r;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                int m4 = acc[mt][nt][0];
#pragma unroll
                for (int j = 1; j < 4; ++j) m4 = acc[mt][nt][j] > m4 ? acc[mt][nt][j] : m4;
                if (!__ballot(!((float)m4 < thr[nt]))) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = (float)acc[mt][nt][j];
                    const uint32_t row = row0 + (uint32_t)mt * 16 + (uint32_t)j;
                    const bool c = !(v < thr[nt]) && row < n_rows32 && q < s.nq;
                    const uint64_t hits = __ballot(c);
                    if (hits) {
                        const uint32_t at = wcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(hits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hits, 0u));
                        if (c && at < s.wcap) {
                            const uint64_t key = make_key(v * qs[nt], row);
                            wl[at] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), q, 0u);
                        }
                        wcount += (uint32_t)__builtin_popcountll(hits);
                    }
                }
            }
        }
    };
    for (uint64_t it = 0; it < my_tiles; ++it) {
        if (it) epilogue(tile_of(blockIdx.x + (it - 1) * gridDim.x));
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (i32x4s){0, 0, 0, 0};
        for (uint32_t kc = 0; kc < nch; ++kc) {
            queries_begin();                              // stage g + 2 -> the slot stage g - 1 was read from (everybody is past that barrier)
            rows_begin();                                 // stage g + 2 -> likewise
            const uint4 *ab = lds + slot * SP3_A_UNITS + a_rd;
            const uint4 *bb = b_lds + bslot * SP_B_UNITS + b_rd;
            i32x4s b0[4], b1[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                b0[nt] = *reinterpret_cast<const i32x4s *>(bb + nt * 128);
                b1[nt] = *reinterpret_cast<const i32x4s *>(bb + nt * 128 + 64);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const i32x4s a0 = *reinterpret_cast<const i32x4s *>(ab + mt * 128);
                const i32x4s a1 = *reinterpret_cast<const i32x4s *>(ab + mt * 128 + 64);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1[nt], acc[mt][nt], 0, 0, 0);
                // the stage's six copy requests, in the order the wait below relies on (queries first), spread over the matrix work
                if (mt == 0) queries_piece(0);
                if (mt == 1) queries_piece(1);
                if (mt == 2) { rows_piece(0); rows_piece(1); }
                if (mt == 3) { rows_piece(2); rows_piece(3); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0[nt], acc[mt][nt], 0, 0, 0);
            }
            slot = slot + 1 == SP3_ARING ? 0 : slot + 1;
            bslot = bslot + 1 == SP8_BRING ? 0 : bslot + 1;
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // all but this stage's six requests have landed: rows and queries of stage g + 1 among them
            sp_stage_barrier();
        }
    }
    epilogue(tile_of(blockIdx.x + (my_tiles - 1) * gridDim.x));
    if (lane == 0) s.wcnt[blockIdx.x * (SP3_THREADS / 64) + (uint32_t)w] = wcount;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // nothing may land in LDS after the block is gone
}

// Other structures of this scan were built in round 4 and measured slower - half stages with 80 KiB of rows in flight per CU (14 % slower: twice the
// barriers), wave-owned rows behind private rings with no stage hand-over (5 % slower), rows straight into the operand registers - and are gone from the
// code since round 6; the measurements stay (profiles/r4_i8_deep_ring.md).

// ---- The same scan with the queries' WHOLE operand image resident in LDS (round 6, last session; the structure scan_sqw.hip found for the scalar-int8 block,
// on this copy's tiled layout).  Rows up to 1 024 coordinates: nch x 16 KiB of query images (96 KiB at 768) + 1 KiB of thresholds and scales, copied once per
// block; after that copy the waves share nothing: no stage barrier, no LDS ring of rows.  A wave OWNS row tiles 2 w and 2 w + 1 of a 256-row tile: their operand
// images are the 4 KiB [4 w KiB, + 4 KiB) of every 32 KiB stage of the copy (sp_unit: ((t * 2 + hl) * 4 + kq) * 16 + swizzle(m) - a lane's 16 bytes sit at one
// lane-constant offset inside each 1 KiB image), so a stage is four `global_load_dwordx4` of 1 KiB each, contiguous, INTO the matrix instruction's operand
// registers, issued AHEAD stages in front of the matrix work and waited for by the kernel's own vmcnt (in-order); per stage and wave 32 matrix instructions
// against all 128 queries and 16 operand reads.  Same integers, same thresholds, same candidates as scan_i8copy_kernel (which stays for longer rows and as option
// i8_resident = 0); LDS per CU 97 KiB instead of 144: every small kernel of the other batches in flight - the sample's exact scores with their 50 KiB query tile
// among them - fits beside a scan. ----
constexpr uint32_t SP8R_MAX_NCH = 8;
__host__ __device__ constexpr size_t sp8r_lds_bytes(uint32_t nch) { return (size_t)nch * SP_B_UNITS * 16 + 2 * SP_QT * 4; }
// (non-temporal: every line of the copy is read once per pass, by one CU, whole - a load instruction covers 1 KiB of consecutive bytes; with plain loads the same
// kernel ran at 0.68 ms per launch instead of 0.61 - 0.63, the step at 90.3 - 90.9 k QPS instead of 96.5 - 97.2 k: profiles/r6_i8_resident.md)
template <int OFF>
__device__ __forceinline__ void sp_gload16(i32x4s &dst, const unsigned char *sbase, uint32_t voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
template <int I, int N, class F>
__device__ __forceinline__ void sp_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sp_static_for<I + 1, N>(f);
    }
}
// Measured and gone from the code (profiles/r6_i8_resident.md): two / three stages of loads ahead (equal or slower: 170 / 200 registers), four row tiles per wave -
// waves 0 - 3 on the block's even tiles, 4 - 7 on its odd ones: half the operand reads per matrix instruction, 250 registers - (equal), a block's tiles consecutive
// instead of every gridDim-th (equal).  Ablation (wrong results, the same launches): without the matrix instructions and operand reads 0.628 ms per launch
// instead of 0.68 - 0.69, with half of them 0.655: the stream alone - this kernel's loads and waits - is 0.76 of HBM, the matrix work costs 9 %.
template <int AHEAD>
__global__ __launch_bounds__(SP3_THREADS, 1) void scan_i8copy_kernel_res(const ScanArgs a, const SplitArgs s) {
    constexpr int SLOTS = AHEAD + 1, WAVES = SP3_THREADS / 64, THREADS = SP3_THREADS, MT = 2, NL = 2 * MT;
    static_assert(SP3_BM == WAVES * MT * 16, "a wave owns two row tiles of a 256-row tile");
    static_assert(NL * AHEAD <= 62, "vmcnt is six bits");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4 *lds = reinterpret_cast<uint4 *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t all_tiles = (a.n_cand + SP3_BM - 1) / SP3_BM;
    const uint64_t n_tiles = split_phase_tiles(all_tiles, s.phase);
    const uint32_t nch = s.nchunks;
    const uint64_t my_tiles = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const uint32_t phase = s.phase & 0xFFu, pstride = split_phase_stride(s.phase);
    auto tile_of = [&](uint64_t j) -> uint64_t { return phase == 0 ? j : phase == 1 ? j * pstride : j + j / (pstride - 1) + 1; };
    if (my_tiles == 0) {
        if (lane == 0) s.wcnt[blockIdx.x * WAVES + (uint32_t)w] = 0;
        return;
    }
    const uint32_t ws = (uint32_t)w;
    auto my_tile = [&](uint64_t i) -> uint64_t { return tile_of(blockIdx.x + i * gridDim.x); };
    float *thr_lds = reinterpret_cast<float *>(smem_raw + (size_t)nch * SP_B_UNITS * 16), *qs_lds = thr_lds + SP_QT;
#pragma unroll 4
    for (uint32_t u = (uint32_t)tid; u < nch * SP_B_UNITS; u += THREADS) lds[u] = s.bq[u];
    for (int u = tid; u < SP_QT; u += THREADS) {
        thr_lds[u] = s.thr[u];
        qs_lds[u] = s.scales[u];
    }
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // from here on the kernel counts its vector-memory traffic itself
    const uint32_t kq_r = (uint32_t)lane >> 4, m_r = (uint32_t)lane & 15u;
    const uint32_t lane_unit = kq_r * 16u + (m_r ^ (2u * kq_r));      // sp_unit(0, 0, kq_r, m_r)
    const uint4 *const b_rd = lds + lane_unit;
    const uint32_t lane_off = lane_unit * 16u;
    auto uniform_ptr = [&](uint64_t v) {
        return reinterpret_cast<const unsigned char *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
                                                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v));
    };
    // one stage's 2 MT loads: the images (row tile MT w + mt, plane hl) of stage kc of a tile -> rc[mt * 2 + hl]
    auto load_stage = [&](uint64_t tile, uint32_t kc, i32x4s (&rc)[NL]) __attribute__((always_inline)) {
        const unsigned char *src = uniform_ptr((uint64_t)(uintptr_t)(s.rows_split + (tile * nch + kc) * SP3_A_UNITS) + ws * (NL * 1024u));
        sp_gload16<0>(rc[0], src, lane_off);
        sp_gload16<1024>(rc[1], src, lane_off);
        sp_gload16<2048>(rc[2], src, lane_off);
        sp_gload16<3072>(rc[3], src, lane_off);
    };
    i32x4s acc[MT][8];
    uint4 *const wl = s.wlist + (uint64_t)(blockIdx.x * WAVES + (uint32_t)w) * s.wcap;
    uint32_t wcount = 0;
    const uint32_t n_rows32 = (uint32_t)a.n_cand;
    // the epilogue of a tile: scan_i8copy_kernel's test (a pair is a candidate unless its accumulator is below the query's threshold), narrowing by
    // wave-uniform steps: the query tile, then the lane's 4 MT rows
    auto epilogue = [&](uint64_t tile) __attribute__((always_inline)) {
        const uint32_t row0 = (uint32_t)(tile * SP3_BM) + ws * (MT * 16u) + 4 * kq_r;
        uint32_t hits8 = 0;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            int mx = acc[0][nt][0];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) mx = acc[mt][nt][j] > mx ? acc[mt][nt][j] : mx;
            if (!((float)mx < thr_lds[nt * 16 + (int)m_r])) hits8 |= 1u << nt;
        }
        if (!__ballot(hits8 != 0)) return;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            if (!__ballot((hits8 >> nt) & 1u)) continue;
            const uint32_t q = (uint32_t)nt * 16 + m_r;
            const float th = thr_lds[q], qs = qs_lds[q];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = (float)acc[mt][nt][j];
                    const uint32_t row = row0 + (uint32_t)mt * 16 + (uint32_t)j;
                    const bool c = !(v < th) && row < n_rows32 && q < s.nq;
                    const uint64_t hits = __ballot(c);
                    if (hits) {
                        const uint32_t at = wcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(hits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hits, 0u));
                        if (c && at < s.wcap) {
                            const uint64_t key = make_key(v * qs, row);
                            wl[at] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), q, 0u);
                        }
                        wcount += (uint32_t)__builtin_popcountll(hits);
                    }
                }
            }
        }
    };
    const uint64_t n_stages = my_tiles * nch;
    // the stage the loads are at, as (tile, kc), by running counters (tile_of divides: once per tile, not per stage); past the block's last stage they
    // repeat it (the waits count loads, not bytes)
    uint64_t itr = 0, tile_r = my_tile(0);
    uint32_t kr = 0;
    auto advance_r = [&]() {
        if (kr + 1 < nch) ++kr;
        else if (itr + 1 < my_tiles) { kr = 0; ++itr; tile_r = my_tile(itr); }
    };
    i32x4s rc[SLOTS][NL];
    sp_static_for<0, AHEAD>([&](auto DC) __attribute__((always_inline)) {
        load_stage(tile_r, kr, rc[decltype(DC)::value]);
        advance_r();
    });
    uint64_t it = 0, tile_c = my_tile(0), tile_done = 0;
    uint32_t kc = 0;
    // one stage; I = g mod SLOTS selects the register sets at compile time
    auto one_stage = [&](auto IC) __attribute__((always_inline)) {
        constexpr int I = decltype(IC)::value, IL = (I + AHEAD) % SLOTS;
        if (kc == 0) {
            if (it) epilogue(tile_done);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) acc[mt][nt] = (i32x4s){0, 0, 0, 0};
        }
        load_stage(tile_r, kr, rc[IL]);                           // stage g + AHEAD -> the register set stage g - 1 ran on
        advance_r();
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NL * AHEAD) : "memory");      // all but the last 2 MT AHEAD loads have landed: this stage's among them
        asm volatile("" : "+v"(rc[I][0]), "+v"(rc[I][1]), "+v"(rc[I][2]), "+v"(rc[I][3]) : : "memory");
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) {
            const uint4 *bb = b_rd + kc * SP_B_UNITS + hl * 64;
            constexpr int RA = 3;
            i32x4s bv[RA + 1];
#pragma unroll
            for (int k = 0; k < RA; ++k) bv[k] = *reinterpret_cast<const i32x4s *>(bb + k * 128);
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
                if (nt + RA < 8) bv[(nt + RA) % (RA + 1)] = *reinterpret_cast<const i32x4s *>(bb + (nt + RA) * 128);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(rc[I][mt * 2 + hl], bv[nt % (RA + 1)], acc[mt][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (++kc == nch) {
            kc = 0;
            tile_done = tile_c;
            if (++it < my_tiles) tile_c = my_tile(it);
        }
    };
    for (uint64_t g = 0; g < n_stages; g += SLOTS) {
        sp_static_for<0, SLOTS>([&](auto IC) __attribute__((always_inline)) {
            if (g + decltype(IC)::value < n_stages) one_stage(IC);
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    epilogue(tile_done);
    if (lane == 0) s.wcnt[blockIdx.x * WAVES + (uint32_t)w] = wcount;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- after a launch: the k best candidates (by approximate score) of every query, for an exact look.  k of them are enough: approximate and exact
// scores differ by a hundredth of the band in practice, so the worst exact score among the k best approximate ones is the k-th best exact score so far
// or next to it - and finding k keys is what the bound-and-rank selection is quick at (the 64 best of ~5 000 took 100 - 190 us, these take ~15) ----
__global__ __launch_bounds__(SEL_BLOCK) void sp_i8_probe_kernel(const uint64_t *cand, const uint32_t *cand_cnt, uint32_t cap, const float *band,
                                                                const int *tile_overflow, uint32_t top, uint32_t *probe_ids, uint32_t *probe_cnt) {
    __shared__ SelShared sel_sh;
    uint64_t (*const sh)[WAVE] = sel_sh.sh;
    SelScratch &scratch = sel_sh.scratch;
    __shared__ uint64_t sh_kth;
    __shared__ uint32_t sh_n;
    const uint32_t q = blockIdx.x;
    const uint32_t raw = cand_cnt[q];
    if (*tile_overflow || raw > cap || raw == 0 || !(band[q] < 3.0e38f)) {       // (sp_select_kernel reports; nothing to learn here)
        if (threadIdx.x == 0) probe_cnt[q] = 0;
        return;
    }
    const uint64_t *c = cand + (uint64_t)q * cap;
    const uint32_t k = top < SP_I8_PROBE ? top : SP_I8_PROBE;
    const uint32_t want = raw < k ? raw : k;
    if (threadIdx.x == 0) sh_n = 0;
    if (raw <= (uint32_t)SEL_SURV) block_kth_key_ranked(c, raw, want, &scratch, &sh_kth);
    else block_kth_key(c, raw, (int)want, sh, &sh_kth);
    const uint64_t kth = sh_kth;                                                  // (keys are distinct - they carry the row -: exactly `want` keys are >= it)
    for (uint32_t base = 0; base < raw; base += SEL_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        if (i < raw) {
            const uint64_t key = c[i];
            if (kth != 0 && key >= kth) {                                          // (`k` is taken: the candidate's key)
                const uint32_t slot = atomicAdd(&sh_n, 1u);
                if (slot < SP_I8_PROBE) probe_ids[(uint64_t)q * SP_I8_PROBE + slot] = key_idx(key);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) probe_cnt[q] = sh_n < SP_I8_PROBE ? sh_n : SP_I8_PROBE;
}
// ... and what their exact scores say: T = the k-th best of them is a lower bound of the final k-th best score; the threshold of the next launch
// and the cut of the selection follow it (never lowered).  One wave per query; the probe list is emptied for the next round.
__global__ __launch_bounds__(64) void sp_i8_bound_kernel(const float *scores, uint32_t *probe_cnt, uint32_t top, const float *band, const float *qscale,
                                                         float *thr, float *t_exact) {
    const uint32_t q = blockIdx.x;
    const int lane = threadIdx.x;
    const uint32_t n = probe_cnt[q];
    const float sc = (uint32_t)lane < n ? scores[(uint64_t)q * SP_I8_PROBE + lane] : -__builtin_inff();
    uint32_t rank = 0;
    for (int j = 0; j < 64; ++j) {
        const float o = __shfl(sc, j, 64);
        rank += (o > sc || (o == sc && j < lane)) ? 1u : 0u;
    }
    if (lane == 0) probe_cnt[q] = 0;
    if (n < top || !(band[q] < 3.0e38f)) return;
    if ((uint32_t)lane < n && rank == top - 1 && sc == sc) {
        if (sc > t_exact[q]) t_exact[q] = sc;
        const float t = (sc - band[q]) / qscale[q];
        if (t > thr[q]) thr[q] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
bool split_scan_ok(const ScanArgs &a) {
    return a.dim % 128 == 0 && a.dim >= 128 && a.rem_pieces == 0 && a.tail_start == a.dim && a.row_stride % 16 == 0 && a.ids == nullptr && a.top <= 64 &&
           !option(OPT_NO_SPLIT_SCAN);
}
size_t split_wlists_counts_bytes(int num_cus) { return ((size_t)num_cus * (SP3_THREADS / 64) * 4 + 255) / 256 * 256; }
size_t split_wlists_bytes(int num_cus) { return split_wlists_counts_bytes(num_cus) + (size_t)num_cus * (SP3_THREADS / 64) * SP_WCAP * 16; }
size_t split_query_bytes(uint32_t dim) { return (size_t)(dim / 32) * SP4_B_UNITS * 16; }   // (sized for the 256-query tile; the half mode needs half of it)

int32_t launch_split_row_stats(hipStream_t st, const void *rows, uint64_t row_stride, uint64_t n, uint32_t dim, uint32_t *d_stats) {
    if (n == 0) return QMX_OK;
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(sp_row_stats_kernel, dim3(2048), dim3(256), 0, st, (const unsigned char *)rows, row_stride, n, dim, d_stats);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}
float split_row_scale(float row_maxabs) {
    if (!(row_maxabs > 0.0f) || !(row_maxabs < 3.0e38f)) return 1.0f;
    int e;
    (void)frexpf(row_maxabs, &e);
    int s = 14 - e;
    s = s > 100 ? 100 : (s < -100 ? -100 : s);
    return ldexpf(1.0f, s);
}

// queries (preprocessed f32, [nq][dim] contiguous) -> bq, qnorm, qmax (per query: the scale of the batch is derived from them where it is needed)
int32_t launch_split_pack_queries(hipStream_t st, const float *d_q, uint32_t nq, uint32_t dim, float *d_qmax, float *d_qnorm, void *d_bq, int half, uint32_t qt) {
    QMX_REQUIRE(nq <= SP_QT_MAX, QMX_ERR_BAD_ARG, "split tile of %u queries", nq);
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(sp_query_stats_kernel, dim3(nq), dim3(256), 0, st, d_q, nq, dim, d_qmax, d_qnorm);
    const uint32_t units = (dim / 32) * qt * 4;
    hipLaunchKernelGGL(sp_pack_queries_kernel, dim3((units + 255) / 256), dim3(256), 0, st, d_q, nq, dim, d_qmax, (uint4 *)d_bq, half, qt);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}
int32_t launch_split_thresholds(hipStream_t st, const uint64_t *d_gthr, const float *d_qnorm, const float *d_qmax, uint32_t nq, float rel_band, float row_norm_max,
                                float row_scale, float *d_scales, float *d_thr, float *d_band, uint32_t qt, uint32_t *d_cand_cnt, uint32_t n_cnt) {
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(sp_thresholds_kernel, dim3(1), dim3(qt), 0, st, d_gthr, d_qnorm, d_qmax, nq, rel_band, row_norm_max, row_scale, d_scales, d_thr, d_band,
                       d_cand_cnt, n_cnt);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}
size_t split_copy_bytes(uint64_t n, uint32_t dim, int half) { return (size_t)((n + SP3_BM - 1) / SP3_BM) * (dim / (half ? 64 : 32)) * SP3_A_UNITS * 16; }
int32_t launch_split_copy(hipStream_t st, const void *rows, uint64_t row_stride, uint64_t n, uint32_t dim, float row_scale, void *d_out, int half) {
    if (n == 0) return QMX_OK;
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(sp_split_copy_kernel, dim3(8192), dim3(256), 0, st, (const unsigned char *)rows, row_stride, n, dim, row_scale, (uint4 *)d_out, half);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

int32_t launch_scan_f32_split(hipStream_t st, const ScanArgs &a, const void *d_bq, float row_scale, const float *d_scales, const float *d_thr,
                              uint64_t *d_cand, uint32_t *d_cand_cnt, uint32_t cap, int num_cus, const void *d_rows_split, int half, void *d_wlists, uint32_t phase,
                              uint32_t qt) {
    auto kfn = scan_f32_split_kernel;
    auto kfn3 = scan_f16pair_kernel<false>;
    auto kfn3h = scan_f16pair_kernel<true>;
    auto kfn4 = scan_f16half256_kernel;
    QMX_REQUIRE(qt == SP_QT || (qt == SP4_QT && half && d_rows_split), QMX_ERR_BAD_ARG, "the 256-query shape scans the half copy");
    static thread_local DeviceOnce attr_once;
    if (attr_once.need()) {
        QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, SP_LDS));
        QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn3), hipFuncAttributeMaxDynamicSharedMemorySize, SP3_LDS));
        QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn3h), hipFuncAttributeMaxDynamicSharedMemorySize, SP3_LDS));
        QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn4), hipFuncAttributeMaxDynamicSharedMemorySize, SP4_LDS));
        attr_once.mark();
    }
    QMX_REQUIRE(!half || d_rows_split, QMX_ERR_BAD_ARG, "the one-product mode scans the half copy only");
    SplitArgs s;
    s.rows_split = (const uint4 *)d_rows_split;
    s.bq = (const uint4 *)d_bq;
    s.nchunks = a.dim / (half ? 64 : 32);
    s.nq = a.nq;
    s.row_scale = row_scale;
    s.scales = d_scales;
    s.thr = d_thr;
    s.cand = d_cand;
    s.cand_cnt = d_cand_cnt;
    s.cap = cap;
    s.wcap = SP_WCAP;
    s.wcnt = (uint32_t *)d_wlists;                                                  // [waves] counts, then the lists (16-byte aligned)
    s.wlist = (uint4 *)((unsigned char *)d_wlists + split_wlists_counts_bytes(num_cus));
    QMX_REQUIRE(!d_rows_split || d_wlists, QMX_ERR_BAD_ARG, "the scan over a derived copy writes per-wave candidate lists");
    QMX_REQUIRE(phase == 0 || d_rows_split, QMX_ERR_BAD_ARG, "phases exist for the scan over a derived copy");
    s.phase = phase;
    const uint32_t bm = qt == SP4_QT ? SP4_BM : d_rows_split ? SP3_BM : SP_BM;
    const uint64_t all_tiles = (a.n_cand + bm - 1) / bm;
    const uint64_t n_tiles = d_rows_split ? split_phase_tiles(all_tiles, phase) : all_tiles;
    if (n_tiles == 0) return QMX_OK;
    const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_tiles, (uint64_t)num_cus));
    ::qmx::clear_stale_error();
    if (qt == SP4_QT) {
        QMX_NOTE_KERNEL(kfn4);
        hipLaunchKernelGGL(kfn4, dim3(grid), dim3(SP3_THREADS), (size_t)SP4_LDS, st, a, s);
    } else if (d_rows_split && half) {
        QMX_NOTE_KERNEL(kfn3h);
        hipLaunchKernelGGL(kfn3h, dim3(grid), dim3(SP3_THREADS), (size_t)SP3_LDS, st, a, s);
    } else if (d_rows_split) {
        QMX_NOTE_KERNEL(kfn3);
        hipLaunchKernelGGL(kfn3, dim3(grid), dim3(SP3_THREADS), (size_t)SP3_LDS, st, a, s);
    } else {
        QMX_NOTE_KERNEL(kfn);
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(SP_THREADS), (size_t)SP_LDS, st, a, s);
    }
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}
// per-wave lists of launch_scan_f32_split (over a derived copy) -> d_cand / d_cand_cnt; d_cand_cnt zeroed by the caller before the scan
int32_t launch_split_regroup(hipStream_t st, const ScanArgs &a, const void *d_wlists, int num_cus, uint64_t *d_cand, uint32_t *d_cand_cnt, uint32_t cap,
                             int *d_overflow, uint32_t phase, uint32_t qt) {
    const uint32_t bm = qt == SP4_QT ? SP4_BM : SP3_BM;
    const uint64_t n_tiles = split_phase_tiles((a.n_cand + bm - 1) / bm, phase);
    if (n_tiles == 0) return QMX_OK;
    const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_tiles, (uint64_t)num_cus));
    const uint32_t n_lists = grid * (SP3_THREADS / 64);
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(sp_regroup_kernel, dim3((n_lists + RG_LISTS - 1) / RG_LISTS), dim3(256), 0, st,
                       (const uint4 *)((const unsigned char *)d_wlists + split_wlists_counts_bytes(num_cus)), (const uint32_t *)d_wlists, (uint32_t)SP_WCAP, n_lists,
                       a.del, d_cand, d_cand_cnt, cap, d_overflow);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}
// any set of per-wave candidate lists (the PQ prefilter's, pq_prefilter.hip) -> per-query lists
int32_t launch_regroup_lists(hipStream_t st, const DeletedView &del, const void *d_wlist, const uint32_t *d_wcnt, uint32_t wcap, uint32_t n_lists, uint64_t *d_cand,
                             uint32_t *d_cand_cnt, uint32_t cap, int *d_overflow) {
    if (n_lists == 0) return QMX_OK;
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(sp_regroup_kernel, dim3((n_lists + RG_LISTS - 1) / RG_LISTS), dim3(256), 0, st, (const uint4 *)d_wlist, d_wcnt, wcap, n_lists, del, d_cand,
                       d_cand_cnt, cap, d_overflow);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}
int32_t launch_split_refine(hipStream_t st, const uint64_t *d_cand, const uint32_t *d_cand_cnt, uint32_t cap, const float *d_band, uint32_t nq, uint32_t top,
                            const float *d_scales, float *d_thr) {
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(sp_refine_kernel, dim3(nq), dim3(SEL_BLOCK), 0, st, d_cand, d_cand_cnt, cap, d_band, top, d_scales, d_thr);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}
int32_t launch_split_select(hipStream_t st, const uint64_t *d_cand, const uint32_t *d_cand_cnt, uint32_t cap, const float *d_band, uint32_t nq, uint32_t top,
                            const VerifyPool &pool, uint32_t q_base, const int *d_tile_overflow, uint32_t *d_ovf_q, SplitStats *d_stats, const float *d_t_exact,
                            bool tighten) {
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(sp_select_kernel, dim3(nq), dim3(SEL_BLOCK), 0, st, d_cand, d_cand_cnt, cap, d_band, top, pool, q_base, d_tile_overflow, d_ovf_q, d_stats,
                       d_t_exact, tighten);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}
int32_t launch_split_plan(hipStream_t st, const uint32_t *d_ovf_q, uint32_t nq, const uint64_t *d_gthr, uint32_t *d_list, uint64_t *d_gthr_packed, uint32_t list_cap,
                          uint32_t *d_count, int *d_run16, int *d_run64, uint32_t n_run64, SplitStats *d_stats, const void *d_queries, uint32_t q_stride,
                          void *d_packed_queries) {
    QMX_REQUIRE(q_stride % 16 == 0, QMX_ERR_OTHER, "query entries are whole 16-byte units");
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(sp_plan_kernel, dim3(1), dim3(256), 0, st, d_ovf_q, nq, d_gthr, d_list, d_gthr_packed, list_cap, d_count, d_run16, d_run64, n_run64, d_stats,
                       (const uint4 *)d_queries, q_stride / 16, (uint4 *)d_packed_queries);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

// ---- the int8 copy ----
bool split_i8_dim_ok(uint32_t dim) { return dim % 128 == 0 && dim >= 128 && dim <= SP_I8_MAX_DIM; }
size_t split_i8_copy_bytes(uint64_t n, uint32_t dim) { return (size_t)((n + SP3_BM - 1) / SP3_BM) * (dim / 128) * SP3_A_UNITS * 16; }
size_t split_i8_query_bytes(uint32_t dim) { return (size_t)(dim / 128) * SP_B_UNITS * 16; }
uint32_t split_i8_probe() { return SP_I8_PROBE; }
// pass 1 of the copy: d_colmax [dim] <- max_r |x_rc| (uint bits), d_colsq [dim] <- sum_r x_rc^2 (both zeroed here)
int32_t launch_split_i8_colstats(hipStream_t st, const void *rows, uint64_t row_stride, uint64_t n, uint32_t dim, uint32_t *d_colmax, float *d_colsq) {
    QMX_REQUIRE(split_i8_dim_ok(dim), QMX_ERR_BAD_ARG, "int8 copy of dim %u", dim);
    ::qmx::clear_stale_error();
    QMX_HIP(hipMemsetAsync(d_colmax, 0, (size_t)dim * 4, st));
    QMX_HIP(hipMemsetAsync(d_colsq, 0, (size_t)dim * 4, st));
    if (n) hipLaunchKernelGGL(sp_i8_colmax_kernel, dim3(2048), dim3(256), 0, st, (const unsigned char *)rows, row_stride, n, dim, d_colmax, d_colsq);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}
// The columns' scales (host, dim numbers).  Any s_i >= max_r |x_ri| / 127 keeps the codes inside [-127, 127] and |e_i| <= 1/2, so the band formula holds for
// every such choice; WHICH choice decides how wide the band is.  With s_i at its floor a block whose columns differ in size (a few dominant coordinates: what
// transformer embeddings with outlier dimensions look like) gives its queries - one scale t per query, set by the largest q_i s_i - codes of 0 / +-1 in
// the small columns: the queries' rounding then costs C2 |f|_2 t on columns whose row codes use the full range.  Raising the small columns' scales to
// 1 / G of the largest typical |q_i s_i| moves range from their row codes (more row error: sum |d_i| / 2 grows) to their query codes (less query error:
// C2 shrinks): the G that minimises the predicted band for queries distributed like the rows is taken.  Predicted band (score units), |q_i| ~ sigma_i:
//   v_i = sigma_i s_i,  t = 3 max v / 127,  rows' rounding 0.4 sum v_i,  queries' rounding t sqrt(sum (sigma_i / s_i)^2) sqrt(sum min(1/12, (v_i / t)^2))
// (restated in tests/test_i8_prefilter_model.py: on eight coordinates twelve times the others the band drops from 1.06 to 0.43 standard deviations of the
// score, the rows inside it from 5 200 to 360 per query; columns of one size - Gaussian, Laplace, Student rows - keep their floor scales: G changes nothing there)
float split_i8_choose_scales(const float *colmax, const float *colsq, uint64_t n, uint32_t dim, float *scale) {
    std::vector<double> s0(dim), sig(dim);
    double wmax = 0.0;
    for (uint32_t i = 0; i < dim; ++i) {
        s0[i] = colmax[i] > 1.0e-30f ? (double)colmax[i] / 127.0 : 1.0;    // (an all-zero or vanishing column: codes 0, |e| = |x| <= 1/2 all the same)
        sig[i] = n ? sqrt((double)colsq[i] / (double)n) : 0.0;
        if (!(sig[i] < 1.0e30)) sig[i] = 0.0;
        wmax = std::max(wmax, sig[i] * s0[i]);
    }
    auto scales_of = [&](double G, std::vector<double> &s) {
        for (uint32_t i = 0; i < dim; ++i) {
            const double w = sig[i] * s0[i];
            s[i] = (w > 0.0 && wmax / G > w && colmax[i] > 1.0e-30f) ? s0[i] * (wmax / G / w) : s0[i];
        }
    };
    auto predicted = [&](const std::vector<double> &s) {
        double vmax = 0.0, vsum = 0.0, c2 = 0.0;
        for (uint32_t i = 0; i < dim; ++i) {
            const double v = sig[i] * s[i];
            vmax = std::max(vmax, v);
            vsum += v;
            c2 += (sig[i] / s[i]) * (sig[i] / s[i]);
        }
        const double t = 3.0 * vmax / 127.0;
        if (!(t > 0.0)) return 0.0;
        double f2 = 0.0;
        for (uint32_t i = 0; i < dim; ++i) f2 += std::min(1.0 / 12.0, (sig[i] * s[i] / t) * (sig[i] * s[i] / t));
        return 0.4 * vsum + t * sqrt(c2) * sqrt(f2);
    };
    static const double grid[] = {1.0e30, 64.0, 45.0, 32.0, 23.0, 16.0, 11.0, 8.0, 5.6, 4.0, 2.8, 2.0};
    std::vector<double> s(dim), best(dim);
    double best_b = 0.0, best_g = grid[0];
    for (size_t k = 0; k < sizeof(grid) / sizeof(grid[0]); ++k) {
        scales_of(grid[k], s);
        const double b = predicted(s);
        if (k == 0 || b < best_b * 0.98) {       // (a candidate has to pay: equal predictions keep the floor scales)
            best_b = b;
            best_g = grid[k];
            best = s;
        }
    }
    for (uint32_t i = 0; i < dim; ++i) {
        float f = (float)best[i];
        const float floor_s = colmax[i] > 1.0e-30f ? colmax[i] / 127.0f : 1.0f;
        scale[i] = f >= floor_s ? f : floor_s;     // (never below the floor, whatever the roundings above did)
    }
    return best_g >= 1.0e29 ? 0.0f : (float)best_g;
}
// pass 2: d_stats [4] <- {C1, C2^2, not finite, -} under the scales d_scale (zeroed here); then the copy itself (split_i8_copy_bytes)
int32_t launch_split_i8_rowstats(hipStream_t st, const void *rows, uint64_t row_stride, uint64_t n, uint32_t dim, const float *d_scale, uint32_t *d_stats) {
    QMX_REQUIRE(split_i8_dim_ok(dim), QMX_ERR_BAD_ARG, "int8 copy of dim %u", dim);
    ::qmx::clear_stale_error();
    QMX_HIP(hipMemsetAsync(d_stats, 0, 16, st));
    if (n) hipLaunchKernelGGL(sp_i8_row_stats_kernel, dim3(2048), dim3(256), 0, st, (const unsigned char *)rows, row_stride, n, dim, d_scale, d_stats);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}
int32_t launch_split_i8_copy(hipStream_t st, const void *rows, uint64_t row_stride, uint64_t n, uint32_t dim, const float *d_scale, void *d_out) {
    if (n == 0) return QMX_OK;
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(sp_i8_copy_kernel, dim3(8192), dim3(256), 0, st, (const unsigned char *)rows, row_stride, n, dim, d_scale, (uint4 *)d_out);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}
// one 128-query tile: codes, scales, bands, first thresholds (d_gthr: the sample's exact k-th best keys); zeroes the tile's candidate counters
int32_t launch_split_i8_pack(hipStream_t st, const float *d_q, uint32_t nq, uint32_t dim, const float *d_scale, const uint64_t *d_gthr, const uint32_t *d_row_stats,
                             float row_norm_max, void *d_bq, float *d_qscale, float *d_band, float *d_thr, float *d_t_exact, uint32_t *d_cand_cnt, uint32_t n_cnt) {
    QMX_REQUIRE(nq <= (uint32_t)SP_QT, QMX_ERR_BAD_ARG, "int8 tile of %u queries", nq);
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(sp_i8_pack_kernel, dim3(SP_QT), dim3(256), (size_t)dim * 4, st, d_q, nq, dim, d_scale, d_gthr, d_row_stats, row_norm_max, (uint4 *)d_bq,
                       d_qscale, d_band, d_thr, d_t_exact, d_cand_cnt, n_cnt);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}
int32_t launch_scan_i8copy(hipStream_t st, const ScanArgs &a, const void *d_bq, const float *d_qscale, const float *d_thr, int num_cus, const void *d_rows_i8,
                           void *d_wlists, uint32_t phase) {
    // rows of up to 1 024 coordinates: the queries' whole image resident in LDS (scan_i8copy_kernel_res; option i8_resident = 0: the staged kernel)
    const uint32_t nch_r = a.dim / 128;
    const bool resident = option(OPT_I8_RESIDENT) > 0 && nch_r <= SP8R_MAX_NCH;
    void (*kfn)(const ScanArgs, const SplitArgs) = resident ? scan_i8copy_kernel_res<1> : scan_i8copy_kernel;
    const size_t lds_bytes = resident ? sp8r_lds_bytes(nch_r) : (size_t)SP8_LDS;
    static thread_local DeviceOnce attr_once;
    if (attr_once.need()) {
        QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(scan_i8copy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SP8_LDS));
        QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(scan_i8copy_kernel_res<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp8r_lds_bytes(SP8R_MAX_NCH)));
        attr_once.mark();
    }
    QMX_REQUIRE(d_rows_i8 && d_wlists && split_i8_dim_ok(a.dim), QMX_ERR_BAD_ARG, "the int8 scan reads the int8 copy and writes per-wave candidate lists");
    SplitArgs s;
    s.rows_split = (const uint4 *)d_rows_i8;
    s.bq = (const uint4 *)d_bq;
    s.nchunks = a.dim / 128;
    s.nq = a.nq;
    s.row_scale = 1.0f;
    s.scales = d_qscale;
    s.thr = d_thr;
    s.cand = nullptr;
    s.cand_cnt = nullptr;
    s.cap = 0;
    s.wcap = SP_WCAP;
    s.wcnt = (uint32_t *)d_wlists;
    s.wlist = (uint4 *)((unsigned char *)d_wlists + split_wlists_counts_bytes(num_cus));
    s.phase = phase;
    const uint64_t n_tiles = split_phase_tiles((a.n_cand + SP3_BM - 1) / SP3_BM, phase);
    if (n_tiles == 0) return QMX_OK;
    const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_tiles, (uint64_t)num_cus));
    ::qmx::clear_stale_error();
    QMX_NOTE_KERNEL(kfn);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(SP3_THREADS), lds_bytes, st, a, s);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}
int32_t launch_split_i8_probe(hipStream_t st, const uint64_t *d_cand, const uint32_t *d_cand_cnt, uint32_t cap, const float *d_band, uint32_t nq, uint32_t top,
                              const int *d_tile_overflow, uint32_t *d_probe_ids, uint32_t *d_probe_cnt) {
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(sp_i8_probe_kernel, dim3(nq), dim3(SEL_BLOCK), 0, st, d_cand, d_cand_cnt, cap, d_band, d_tile_overflow, top, d_probe_ids, d_probe_cnt);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}
int32_t launch_split_i8_bound(hipStream_t st, const float *d_scores, uint32_t *d_probe_cnt, uint32_t nq, uint32_t top, const float *d_band, const float *d_qscale,
                              float *d_thr, float *d_t_exact) {
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(sp_i8_bound_kernel, dim3(nq), dim3(64), 0, st, d_scores, d_probe_cnt, top, d_band, d_qscale, d_thr, d_t_exact);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

}  // namespace qmx

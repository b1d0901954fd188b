// scan_split.hip — f32 dot / cosine brute-force top-k for 65..128 queries per pass: a matrix-core PREFILTER with exact verification.
//
// Why.  The exact chain-major scan (scan_mfma16.hip) reproduces dot_similarity_avx bit for bit on v_mfma_f32_16x16x4_f32, which runs
// at the vector-ALU rate (155 TFLOP/s measured, profiles/r2_mfma_issue_rates.txt): 64 queries x 10 M x 768 cost 6.3 ms of matrix time at
// peak against a 4.5 ms HBM stream, and every further query costs another 0.1 ms.  The f16 matrix instruction is 16x faster.  An f32 value
// splits EXACTLY into two f16 values plus a residual below 2^-22 of it (x 2^e = h + l + r, h = f16(x 2^e), l = f16(x 2^e - h)), so
//     sum x_i y_i  =  2^-(ex + ey) [ sum h h' + sum h l' + sum l h' ]  +  O(2^-21) sum |x_i y_i|
// three v_mfma_f32_16x16x32_f16 per 16 x 16 x 32 block, f32 accumulation.  That approximate score is NOT returned to anybody: it only
// decides which rows are worth an exact look.
//   1. prescan   (api.hip)  exact scores of a strided sample of the block -> T_q = exact k-th best of the sample <= final k-th best
//   2. this file            approximate score A(r, q) of EVERY row; rows with A >= T_q - b_q become candidates (~1000 k per query)
//   3. select               A_k = k-th best approximate score among the candidates; keep those with A >= A_k - 2 b_q  (~k rows)
//   4. verify    (api.hip)  exact scores of the kept rows with the gather kernel of qmx_rescore (the reference's bits), sort, top k
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
#include "scan_common.hpp"

namespace qmx {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4s __attribute__((ext_vector_type(4)));

constexpr int SP_BM = 256;            // rows per tile
constexpr int SP_QT = 128;            // queries per pass
constexpr int SP_THREADS = 512;
constexpr int SP_A_UNITS = SP_BM * 2 * 4;     // 16-byte units of one A chunk buffer (256 rows x {h, l} x 4 k-groups) = 32 KB
constexpr int SP_B_UNITS = SP_QT * 2 * 4;     // ... of one B chunk buffer = 16 KB
constexpr int SP_LDS = (2 * SP_A_UNITS + 2 * SP_B_UNITS) * 16;

// unit index of (16-row or 16-query tile t, half hl, k-group kq, row-in-tile m) inside a chunk buffer
__device__ __forceinline__ uint32_t sp_unit(uint32_t t, uint32_t hl, uint32_t kq, uint32_t m) { return ((t * 2 + hl) * 4 + kq) * 16 + (m ^ (2 * kq)); }

// x * scale = h + l (+ a residual below 2^-22 |x * scale|); round to nearest even both times
__device__ __forceinline__ void sp_split4(const f32x4s v, float scale, half4 &h, half4 &l) {
    const float x[4] = {v[0] * scale, v[1] * scale, v[2] * scale, v[3] * scale};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = (_Float16)x[i];
        l[i] = (_Float16)(x[i] - (float)h[i]);
    }
}

// ---- once per query batch: preprocessed f32 queries -> split f16 in the B-operand layout, per-query norms, the scale of the batch ----
// stats[0] = max |q| over the batch (uint bits of a non-negative float order like the float)
__global__ void sp_query_stats_kernel(const float *q, uint32_t nq, uint32_t dim, uint32_t *stats, float *qnorm) {
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
        atomicMax(stats, __float_as_uint(smx[0]));
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
__global__ void sp_scales_kernel(const uint32_t *stats, float row_scale, float *scales) {
    const float qs = sp_pow2_scale(__uint_as_float(stats[0]));
    scales[0] = qs;
    scales[1] = row_scale * qs;
    scales[2] = 1.0f / (row_scale * qs);
}
// one thread per 16-byte unit: bq[kc][nt][hl][kq][n ^ 2 kq] = 8 halfs, k = 32 kc + 8 kq + e, query 16 nt + n (zero beyond nq)
__global__ void sp_pack_queries_kernel(const float *q, uint32_t nq, uint32_t dim, const float *scales, uint4 *bq) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nchunks = dim / 32;
    if (gid >= nchunks * SP_QT * 4) return;
    const uint32_t kc = gid / (SP_QT * 4), r = gid % (SP_QT * 4);
    const uint32_t nt = r / 64, kq = (r / 16) % 4, n = r % 16;
    const uint32_t qi = nt * 16 + n;
    const float scale = scales[0];
    half8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = qi < nq ? q[(uint64_t)qi * dim + kc * 32 + kq * 8 + e] * scale : 0.0f;
        h[e] = (_Float16)x;
        l[e] = (_Float16)(x - (float)h[e]);
    }
    uint4 *chunk = bq + (uint64_t)kc * SP_B_UNITS;
    chunk[sp_unit(nt, 0, kq, n)] = *reinterpret_cast<const uint4 *>(&h);
    chunk[sp_unit(nt, 1, kq, n)] = *reinterpret_cast<const uint4 *>(&l);
}
// thr[q] in accumulator units: (exact k-th best of the sample - band) * scale; -inf when the sample has no k-th best (everything is a candidate,
// the buffers overflow, the exact scan takes over).  band[q] = rel_band * row_norm_max * |q| (score units).
__global__ void sp_thresholds_kernel(const uint64_t *gthr, const float *qnorm, uint32_t nq, float rel_band, float row_norm_max, const float *scales,
                                     float *thr, float *band) {
    const uint32_t q = threadIdx.x;
    if (q >= SP_QT) return;
    if (q >= nq) { thr[q] = __builtin_inff(); band[q] = 0.0f; return; }
    const float b = rel_band * row_norm_max * qnorm[q];
    band[q] = b;
    const uint64_t k = gthr[q];
    thr[q] = k ? (key_score(k) - b) * scales[1] : -__builtin_inff();
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

struct SplitArgs {
    const uint4 *bq;        // split queries, [dim / 32][SP_B_UNITS]
    uint32_t nchunks;       // dim / 32
    uint32_t nq;            // live queries (<= 128)
    float row_scale;        // power of two applied to the rows before the split
    const float *scales;    // device: [1] = accumulator units per score unit, [2] = inverse
    const float *thr;       // [128] candidate threshold, accumulator units
    uint64_t *cand;         // [128][cap] keys (approximate score, row)
    uint32_t *cand_cnt;     // [128] appended (may run past cap: overflow)
    uint32_t cap;
};

__global__ __launch_bounds__(SP_THREADS, 1) __attribute__((amdgpu_waves_per_eu(1, 2))) void scan_f32_split_kernel(const ScanArgs a, const SplitArgs s) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint4 *lds = reinterpret_cast<uint4 *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wm = (uint32_t)w & 3u, wn = (uint32_t)w >> 2;
    const unsigned char *rows = reinterpret_cast<const unsigned char *>(a.rows);
    const uint64_t n_tiles = (a.n_cand + SP_BM - 1) / SP_BM;
    const uint32_t nch = s.nchunks;
    const float rscale = s.row_scale;

    // loader role: rows 32 w + 8 j + (lane >> 3), j = 0..3, 16-byte piece p = lane & 7 of the chunk's 128 bytes
    const uint32_t p = (uint32_t)lane & 7u, lrow0 = (uint32_t)w * 32u + ((uint32_t)lane >> 3);
    const uint32_t kq_w = p >> 1;                      // k-group of the piece
    uint32_t a_wr[4];                                  // byte offsets of this thread's 8-byte stores inside an A buffer (h half; l is + 64 units)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t rl = lrow0 + 8u * (uint32_t)j;
        a_wr[j] = sp_unit(rl >> 4, 0, kq_w, rl & 15u) * 16u + (p & 1u) * 8u;
    }
    // MFMA role: A units of row tile 4 wm + mt, B units of query tile 4 wn + nt; k-group lane >> 4, row / query lane & 15
    const uint32_t kq_r = (uint32_t)lane >> 4, m_r = (uint32_t)lane & 15u;
    const uint32_t a_rd = sp_unit(wm * 4, 0, kq_r, m_r), b_rd = sp_unit(wn * 4, 0, kq_r, m_r);   // + 128 units per tile, + 64 for the l half

    float thr[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) thr[nt] = s.thr[wn * 64 + nt * 16 + m_r];
    const float inv_scale = s.scales[2];

    f32x4s areg[4];
    const unsigned char *rowp[4];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(sp_lds_byte *)smem_raw;
    const uint32_t lane_off = (uint32_t)lane * 16u;
    auto set_tile = [&](uint64_t tile) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint64_t r = tile * SP_BM + lrow0 + 8u * (uint32_t)j;
            if (r >= a.n_cand) r = a.n_cand - 1;         // rows past the end: any valid row, results masked
            rowp[j] = rows + r * a.row_stride + p * 16u;
        }
    };
    // rows of chunk kc -> registers; the queries' chunk kc -> LDS buffer `bbuf` directly (this wave's 2 KiB of its 16 KiB)
    auto load_chunk = [&](uint32_t kc, uint32_t bbuf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) areg[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4s *>(rowp[j] + kc * 128u));
        const unsigned char *bsrc = reinterpret_cast<const unsigned char *>(s.bq + (uint64_t)kc * SP_B_UNITS) + (uint32_t)w * 2048u;
        const uint32_t dst = lds0 + (2u * SP_A_UNITS + bbuf * SP_B_UNITS) * 16u + (uint32_t)w * 2048u;
        // (wave-uniform base through SGPRs; readfirstlane returns a signed int: widen the halves as unsigned)
        const uint64_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)bsrc);
        const uint64_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uintptr_t)bsrc >> 32));
        const unsigned char *ub = reinterpret_cast<const unsigned char *>((hi << 32) | lo);
        sp_glds16(ub, lane_off, dst);
        sp_glds16(ub + 1024u, lane_off, dst + 1024u);
    };
    auto store_chunk = [&](uint32_t buf) {
        unsigned char *ab = reinterpret_cast<unsigned char *>(lds + buf * SP_A_UNITS);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            half4 h, l;
            sp_split4(areg[j], rscale, h, l);
            *reinterpret_cast<half4 *>(ab + a_wr[j]) = h;
            *reinterpret_cast<half4 *>(ab + a_wr[j] + 64 * 16) = l;
        }
    };

    uint32_t buf = 0;
    uint64_t tile = blockIdx.x;
    if (tile < n_tiles) {
        set_tile(tile);
        load_chunk(0, 0);
    }
    for (; tile < n_tiles; tile += gridDim.x) {
        f32x4s acc[4][4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4s){0.f, 0.f, 0.f, 0.f};
        for (uint32_t kc = 0; kc < nch; ++kc) {
            store_chunk(buf);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the queries' chunk (LDS-DMA, not counted by the compiler) has landed
            __syncthreads();
            // the next chunk (of this tile or of the block's next one) travels while this one is multiplied
            if (kc + 1 < nch) {
                load_chunk(kc + 1, buf ^ 1);
            } else if (tile + gridDim.x < n_tiles) {
                set_tile(tile + gridDim.x);
                load_chunk(0, buf ^ 1);
            }
            const uint4 *ab = lds + buf * SP_A_UNITS + a_rd;
            const uint4 *bb = lds + 2 * SP_A_UNITS + buf * SP_B_UNITS + b_rd;
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
                for (int nt = 0; nt < 4; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[nt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[nt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[nt], acc[mt][nt], 0, 0, 0);
                }
            }
            buf ^= 1;
        }
        // ---- candidates of the tile: accumulator register j of (mt, nt) = row 64 wm + 16 mt + 4 (lane >> 4) + j, query 64 wn + 16 nt + (lane & 15)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const uint32_t q = wn * 64 + (uint32_t)nt * 16 + m_r;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = acc[mt][nt][j];
                    const uint64_t row = tile * SP_BM + wm * 64 + (uint32_t)mt * 16 + 4 * kq_r + (uint32_t)j;
                    const bool c = !(v < thr[nt]) && row < a.n_cand && q < s.nq;     // NaN (greatest in OrderedFloat) is a candidate
                    if (__ballot(c)) {
                        if (c && a.del.live((uint32_t)row)) {
                            const uint32_t slot = atomicAdd(&s.cand_cnt[q], 1u);
                            if (slot < s.cap) s.cand[(uint64_t)q * s.cap + slot] = make_key(v * inv_scale, (uint32_t)row);
                        }
                    }
                }
            }
    }
}

// ---- candidates -> the rows worth an exact score: A_k = k-th best approximate key, keep A >= A_k - 2 band (at most vcap per query) ----
constexpr int SEL_BLOCK = 512;
__global__ __launch_bounds__(SEL_BLOCK) void sp_select_kernel(const uint64_t *cand, const uint32_t *cand_cnt, uint32_t cap, const float *band, uint32_t top,
                                                              uint32_t vcap, uint32_t *ver_ids, uint32_t *ver_cnt, int *overflow) {
    __shared__ uint64_t sh[SEL_BLOCK / WAVE][WAVE];
    __shared__ float sh_cut;
    __shared__ uint32_t sh_n;
    const uint32_t q = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t raw = cand_cnt[q];
    if (raw > cap) {                          // the candidate buffer overflowed: this pass cannot be trusted
        if (threadIdx.x == 0) { *overflow = 1; ver_cnt[q] = 0; }
        return;
    }
    const uint64_t *c = cand + (uint64_t)q * cap;
    const int ptop = (int)top;
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
    if (threadIdx.x == 0) sh_n = 0;
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
        const uint64_t kth = readlane_u64(merged, ptop - 1);
        // fewer than k candidates: keep all of them (cut = -inf)
        if (lane == 0) sh_cut = kth ? key_score(kth) - 2.0f * band[q] : -__builtin_inff();
    }
    __syncthreads();
    const float cut = sh_cut;
    for (uint32_t base = 0; base < raw; base += SEL_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        if (i < raw) {
            const uint64_t key = c[i];
            if (!(key_score(key) < cut)) {
                const uint32_t slot = atomicAdd(&sh_n, 1u);
                if (slot < vcap) ver_ids[(uint64_t)q * vcap + slot] = key_idx(key);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (sh_n > vcap) { *overflow = 1; ver_cnt[q] = 0; }
        else ver_cnt[q] = sh_n;
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

// ---------------------------------------------------------------------------------------------------------------------------
bool split_scan_ok(const ScanArgs &a) {
    return a.dim % 32 == 0 && a.dim >= 32 && a.rem_pieces == 0 && a.tail_start == a.dim && a.row_stride % 16 == 0 && a.ids == nullptr && a.top <= 64 &&
           !option(OPT_NO_SPLIT_SCAN);
}
size_t split_query_bytes(uint32_t dim) { return (size_t)(dim / 32) * SP_B_UNITS * 16; }

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

// queries (preprocessed f32, [nq][dim] contiguous) -> bq, qnorm, scales.  d_stats: one zeroed u32.
int32_t launch_split_pack_queries(hipStream_t st, const float *d_q, uint32_t nq, uint32_t dim, float row_scale, uint32_t *d_stats, float *d_qnorm,
                                  float *d_scales, void *d_bq) {
    ::qmx::clear_stale_error();
    QMX_HIP(hipMemsetAsync(d_stats, 0, 4, st));
    hipLaunchKernelGGL(sp_query_stats_kernel, dim3(nq), dim3(256), 0, st, d_q, nq, dim, d_stats, d_qnorm);
    hipLaunchKernelGGL(sp_scales_kernel, dim3(1), dim3(1), 0, st, d_stats, row_scale, d_scales);
    const uint32_t units = (dim / 32) * SP_QT * 4;
    hipLaunchKernelGGL(sp_pack_queries_kernel, dim3((units + 255) / 256), dim3(256), 0, st, d_q, nq, dim, d_scales, (uint4 *)d_bq);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}
int32_t launch_split_thresholds(hipStream_t st, const uint64_t *d_gthr, const float *d_qnorm, uint32_t nq, float rel_band, float row_norm_max,
                                const float *d_scales, float *d_thr, float *d_band) {
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(sp_thresholds_kernel, dim3(1), dim3(SP_QT), 0, st, d_gthr, d_qnorm, nq, rel_band, row_norm_max, d_scales, d_thr, d_band);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}
int32_t launch_scan_f32_split(hipStream_t st, const ScanArgs &a, const void *d_bq, float row_scale, const float *d_scales, const float *d_thr,
                              uint64_t *d_cand, uint32_t *d_cand_cnt, uint32_t cap, int num_cus) {
    auto kfn = scan_f32_split_kernel;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, SP_LDS));
        attr_set = true;
    }
    SplitArgs s;
    s.bq = (const uint4 *)d_bq;
    s.nchunks = a.dim / 32;
    s.nq = a.nq;
    s.row_scale = row_scale;
    s.scales = d_scales;
    s.thr = d_thr;
    s.cand = d_cand;
    s.cand_cnt = d_cand_cnt;
    s.cap = cap;
    const uint64_t n_tiles = (a.n_cand + SP_BM - 1) / SP_BM;
    const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_tiles, (uint64_t)num_cus));
    ::qmx::clear_stale_error();
    QMX_NOTE_KERNEL(kfn);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(SP_THREADS), (size_t)SP_LDS, st, a, s);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}
int32_t launch_split_select(hipStream_t st, const uint64_t *d_cand, const uint32_t *d_cand_cnt, uint32_t cap, const float *d_band, uint32_t nq, uint32_t top,
                            uint32_t vcap, uint32_t *d_ver_ids, uint32_t *d_ver_cnt, int *d_overflow) {
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(sp_select_kernel, dim3(nq), dim3(SEL_BLOCK), 0, st, d_cand, d_cand_cnt, cap, d_band, top, vcap, d_ver_ids, d_ver_cnt, d_overflow);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

}  // namespace qmx

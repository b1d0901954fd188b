// pq_prefilter.hip — brute-force top-k over PQ codes for 4 and more queries: an integer PREFILTER with exact verification.
//
// Why.  `EncodedVectorsPQ::score_point` (lib/quantization/src/encoded_vectors_pq.rs:409-493) is m table gathers per (row, query):
// 96 random 4-byte reads of a 96 KiB LUT at m = 96.  With one lane per row and the f32 LUT of ONE query in LDS (pq.hip, pq_scan_kernel)
// the gathers of a wave land on random banks: counters say 68 % of the LDS cycles are conflicts, and a 32-query scan of 10 M rows takes
// 2 x 3.2 ms for 0.96 GB of codes (0.04 of the HBM stream).  The f32 sums cannot be reordered (score_point_sse's four lane accumulators
// are sequential in chunk order: the bits), so the exact kernel cannot be laid out conflict-free.  An approximate integer score can:
//   * the LUT of a query is quantised to 8 bits per entry with ONE step for the whole query: LUT[c][j] = lo_c + step (q_cj + d), |d| <= 1/2,
//     so   exact score = sum_c lo_c + step (A + D),  A = sum_c q_{c, code_c} (an integer), |D| <= m / 2: a rigorous band around A;
//   * four queries share a dword (one byte each): ONE ds_read_b32 per (row, chunk) serves four queries;
//   * conflict-free by construction: the LDS table is [code][slot], slot = chunk (+ chunks 0..30 once more behind the last), and lane i
//     of a 32-lane group visits the chunks of ITS row in the rotated order i, i + 1, i + 2, ... (mod m_pad): at every step the 32 lanes
//     of a group read 32 consecutive slots = 32 different banks, whatever their codes (MI355X guide, LDS table: ds_read_b32 is served in
//     two groups of 32 lanes over 32 banks).  The rotation lives in the data: a derived copy of the code block (`pq_rotate_kernel`, built
//     once per segment, m_pad bytes per row) stores byte t of row r as the code of chunk (t + r mod 32) mod m_pad, so a lane's code bytes
//     sit at fixed register positions and the gather address is code * stride + 4 i (+ 4 t as the instruction's immediate offset);
//   * the byte sums cost no vector instruction: the four dwords a lane gathers for four consecutive chunks ARE the A operand of one
//     v_mfma_i32_16x16x64_i8 (16 bytes per lane), and a constant 0 / 1 B operand routes byte k of lane group g to output column 4 g + k:
//     D[m][4 g + k] += sum of the bytes of query k in the 4 dwords of lane m + 16 g.  The matrix core is used as a 64-lane byte-unpacking
//     adder with i32 accumulators (entries are stored less 128: the operand is signed), which leaves two vector instructions per gather
//     (code extraction, address) - the integer ALU rate (one wave instruction per clock and CU) is what bounds this kernel.
// The approximate score is never returned.  Like the f32 prefilter (scan_split.hip): a strided sample scored exactly gives T_q <= the final
// k-th best score; rows with A >= (T_q - L) / step - band become candidates (per-wave lists, plain stores); `sp_select_kernel` keeps those
// within two bands of the k-th best approximate score; `pq_pair_kernel` re-scores them in the reference's order and `sort_scored_kernel`
// returns the top k: ids, score bits and tie order of the exact scan.  A query whose lists overflow (or whose LUT has non-finite entries)
// takes the exact scan, alone (api_*.hip).
//
// Roofline: the stream is m_pad bytes per row and 4-query group (through L2 for all groups but the first: blocks of one row slab and
// different query groups are placed on the same XCD, see `pqf_block_role`); what binds from 8 queries up is LDS issue (2 cycles per
// ds_read_b32) and the two vector instructions per gather.
#include <algorithm>
#include <atomic>

#include "kernels.hpp"

namespace qmx {

constexpr int PQF_THREADS = 1024;
constexpr int PQF_WAVES = PQF_THREADS / 64;
// rounding of one chunk's quantised entry, in table units: 1/2 from rint, plus the f32 evaluation of (v - lo) * inv_step - the subtraction, the rounded
// inv_step and the product each within 2^-24 relative of a value of at most PQF_QMAX: 3 * PQF_QMAX * 2^-24 = 4.6e-5 at 255 - rounded up to 1e-4.  Tied to
// PQF_QMAX: a wider table needs a wider constant.
constexpr int PQF_QMAX = 255;                     // quantised LUT entries 0..255, stored less 128 (the matrix core's int8 operand is signed)
constexpr double PQF_ROUND = 0.5 + 1.0e-4;
static_assert(3.0 * PQF_QMAX * 5.9604644775390625e-08 < 1.0e-4, "PQF_ROUND must cover the f32 rounding of a quantised LUT entry");
typedef int pqf_i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const int pqf_lds_int;
typedef __attribute__((address_space(3))) unsigned char pqf_lds_byte;
// dwords per code row of the LDS table: the chunks, chunks 0..31 once more behind them, rounded up to a power of two (the gather address is a shift + or)
__host__ __device__ constexpr uint32_t pqf_slots(uint32_t m_pad) { return m_pad <= 32 ? 64u : 128u; }

// ---- the rotated copy: out[(T * NP + p) * 32 + i] = 16 code bytes of row 32 T + i: byte e = code of chunk (16 p + e + i) mod m_pad
// (0 for chunks >= m and rows >= n: their table entries are 0) ----
__global__ __launch_bounds__(256) void pq_rotate_kernel(const uint8_t *codes, uint64_t row_stride, uint64_t n, uint32_t m, uint32_t m_pad, uint64_t n_pad,
                                                        uint4 *out) {
    const uint32_t np = m_pad / 16;
    for (uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x; gid < n_pad * np; gid += (uint64_t)gridDim.x * 256) {
        const uint64_t T = gid / (32ull * np);
        const uint32_t rem = (uint32_t)(gid % (32ull * np)), p = rem / 32, i = rem % 32;
        const uint64_t r = T * 32 + i;
        uint32_t w[4] = {0, 0, 0, 0};
        if (r < n) {
            const uint8_t *row = codes + r * row_stride;
#pragma unroll
            for (uint32_t e = 0; e < 16; ++e) {
                const uint32_t c = (16 * p + e + i) % m_pad;
                const uint32_t v = c < m ? row[c] : 0u;
                w[e / 4] |= v << (8 * (e % 4));
            }
        }
        out[gid] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
// (round 4 also kept this copy with every code in 16 bits - one vector instruction per gather address instead of two, twice the bytes: the same 0.96 ms at
// 32 queries, twice the time at 4 - gone from the code since round 6, profiles/r4_pq_prefilter_w16.md)
size_t pq_rot_bytes(uint64_t n, uint32_t m) {
    const uint32_t m_pad = (m + 31) / 32 * 32;
    return (size_t)((n + 63) / 64 * 64) * m_pad;
}
bool pq_prefilter_shape_ok(uint32_t m, uint32_t ncent) { return m >= 1 && m <= 96 && ncent >= 1 && ncent <= 256; }
int32_t launch_pq_rotate(hipStream_t st, const void *codes, uint64_t row_stride, uint64_t n, uint32_t m, void *d_out) {
    if (n == 0) return QMX_OK;
    const uint32_t m_pad = (m + 31) / 32 * 32;
    const uint64_t n_pad = (n + 63) / 64 * 64;
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(pq_rotate_kernel, dim3(4096), dim3(256), 0, st, (const uint8_t *)codes, row_stride, n, m, m_pad, n_pad, (uint4 *)d_out);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

// ---- per query: the 8-bit table, the candidate threshold and the band ----
// One block per query, thread j = centroid j.  lut = the query's f32 LUT [m][ncent] (EncodedQueryPQ, encode_query :519-541).
//   lo_c = min_j LUT[c][j], R = max_c (max_j - min_j), step = R / 255, q_cj = rint((LUT[c][j] - lo_c) / step)  in 0..255
//   exact score (real arithmetic) = L + step (A + D), L = sum_c lo_c, |D| <= PQF_ROUND m  (0.5 per chunk + the f32 rounding of the quotient: see PQF_ROUND)
//   the f32 score the exact kernels return differs from the real sum by at most E = (m + 1) 2^-24 sum_c max_j |LUT[c][j]|
//   => a row with exact score >= T has A >= (T - L - E) / step - PQF_ROUND m =: thr (floored, minus 1);   band (A units) = PQF_ROUND m + E / step + 1
// table8: bytes (q - 128 as int8), [group][code][slots] dwords, byte k of a dword = query 4 group + k; slots >= m_pad repeat chunks 0..31.  The
// caller fills the table with 0x80 (= 0) first: padding chunks, missing centroids and the unused query bytes of the last group must read 0.
__global__ __launch_bounds__(256) void pq_lut8_kernel(const unsigned char *luts, uint32_t q_stride, uint32_t nq, uint32_t m, uint32_t ncent, uint32_t m_pad,
                                                      const uint64_t *gthr, uint8_t *table8, int32_t *thr, float *band) {
    __shared__ float sh_lo[128], sh_hi[128], sh_ab[128];
    __shared__ int sh_bad;
    __shared__ float sh_R, sh_E;
    __shared__ double sh_L;
    const uint32_t q = blockIdx.x, j = threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *lut = reinterpret_cast<const float *>(luts + (uint64_t)q * q_stride);
    const uint32_t slots = pqf_slots(m_pad);
    if (j == 0) sh_bad = 0;
    __syncthreads();
    // per chunk: min, max, max |.| of its centroids' entries - a WAVE per chunk (chunks wave, wave + 4, ...: no block barrier inside the loop, and the
    // loads of the wave's next chunks are in flight while one is reduced; the first version synchronised the block twice per chunk: 100 us per launch)
    int bad = 0;
    for (uint32_t c = (uint32_t)wave; c < m; c += 4) {
        float mn = __builtin_inff(), mx = -__builtin_inff(), ab = 0.0f;
        for (uint32_t jj = (uint32_t)lane; jj < ncent; jj += 64) {
            const float v = lut[(uint64_t)c * ncent + jj];
            const bool fin = !(v != v) && __builtin_fabsf(v) < 3.0e38f;
            if (!fin) bad = 1;
            mn = __builtin_fminf(mn, v);
            mx = __builtin_fmaxf(mx, v);
            ab = __builtin_fmaxf(ab, __builtin_fabsf(v));
        }
        for (int o = 32; o >= 1; o >>= 1) {
            mn = __builtin_fminf(mn, __shfl_xor(mn, o, 64));
            mx = __builtin_fmaxf(mx, __shfl_xor(mx, o, 64));
            ab = __builtin_fmaxf(ab, __shfl_xor(ab, o, 64));
        }
        if (lane == 0) { sh_lo[c] = mn; sh_hi[c] = mx; sh_ab[c] = ab; }
    }
    if (bad) sh_bad = 1;
    __syncthreads();
    if (j == 0) {           // the sums in chunk order, as before
        float R0 = 0.0f, E0 = 0.0f;
        double L0 = 0.0;
        for (uint32_t c = 0; c < m; ++c) {
            R0 = __builtin_fmaxf(R0, sh_hi[c] - sh_lo[c]);
            E0 += sh_ab[c];
            L0 += (double)sh_lo[c];
        }
        sh_R = R0; sh_E = E0; sh_L = L0;
    }
    __syncthreads();
    const float R = sh_R, E = sh_E;
    const double L = sh_L;
    bad = sh_bad;
    // degenerate tables (every entry equal, or non-finite entries): all-zero table, see `usable` below
    const bool flat = !(R > 0.0f) || !(R < 3.0e38f) || bad;
    const float inv_step = flat ? 0.0f : (float)PQF_QMAX / R;
    const uint32_t group = q / 4, k = q % 4;
    uint8_t *tab = table8 + ((uint64_t)group * 256 + j) * slots * 4 + k;
    if (j < ncent) {
        // eight entries per trip: their loads go out together (a byte store may alias anything, so the compiler keeps every load of a plain loop behind
        // the stores of the iteration before it: 96 dependent round trips)
        for (uint32_t c0 = 0; c0 < m; c0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = c0 + u < m ? lut[(uint64_t)(c0 + u) * ncent + j] : 0.0f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t c = c0 + (uint32_t)u;
                if (c < m) {
                    float x = __builtin_rintf((v[u] - sh_lo[c]) * inv_step);
                    x = __builtin_fminf(__builtin_fmaxf(x, 0.0f), (float)PQF_QMAX);
                    const uint8_t b = (uint8_t)((flat ? 0u : (uint32_t)x) ^ 0x80u);
                    tab[(uint64_t)c * 4] = b;
                    if (c < 32) tab[(uint64_t)(m_pad + c) * 4] = b;
                }
            }
        }
    }
    if (j == 0) {
        const double step = flat ? 1.0 : (double)R / (double)PQF_QMAX;
        const double Es = (double)(m + 1) * 5.9604644775390625e-08 * (double)E;      // 2^-24
        const double bnd = PQF_ROUND * (double)m + Es / step + 1.0;
        const uint64_t key = gthr[q];
        // no sample bound (fewer than k live rows in the sample) or a degenerate table: no candidates, and the infinite band tells
        // sp_select_kernel that this query takes the exact scan
        const bool usable = key != 0 && !flat;
        double t = usable ? ((double)key_score(key) - L - Es) / step - PQF_ROUND * (double)m - 1.0 : 0.0;
        t = __builtin_floor(t);
        thr[q] = !usable ? 0x7F7F7F7F : (t < 0.0 ? 0 : (t > 1.0e6 ? 1000000 : (int32_t)t));
        band[q] = usable ? (float)bnd : __builtin_inff();
    }
}

struct PqfArgs {
    const uint4 *rot;         // the rotated copy of the code block
    const uint32_t *table8;   // [n_groups][256][slots] dwords
    const int32_t *thr;       // [4 n_groups] candidate iff A >= thr (unused query slots: INT_MAX)
    uint32_t n_groups, n_slabs;
    uint4 *wlist;             // [waves][wcap] (key lo = ~id, key hi = ord(A), query of the tile, 0)
    uint32_t *wcnt;           // [waves] entries each wave wanted to append (may exceed wcap: overflow)
    uint32_t wcap;
};

// block -> (row slab, query group).  Blocks are dealt to the 8 XCDs round-robin (block b runs on XCD b mod 8), each XCD has its own L2: the
// n_groups blocks that stream the SAME rows for different query groups get consecutive places on ONE XCD, so that all but the first find the
// rows in that XCD's L2.  n_slabs is a multiple of 8.
__device__ __forceinline__ void pqf_block_role(uint32_t b, uint32_t n_groups, uint32_t &slab, uint32_t &group) {
    const uint32_t xcd = b % 8, k = b / 8;
    group = k % n_groups;
    slab = (k / n_groups) * 8 + xcd;
}

template <int NP /* 16-byte pieces of a rotated row of 8-bit codes: m_pad = 16 NP */>
__global__ __launch_bounds__(PQF_THREADS, 1) void pq_prefilter_kernel(const ScanArgs a, const PqfArgs f) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t M_PAD = 16 * NP, SLOTS = pqf_slots(M_PAD), SH = SLOTS == 64 ? 8 : 9;      // a code row = SLOTS dwords = 1 << SH bytes
    constexpr int NPW = NP;                                                                        // pieces of a row in the copy
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t slab, group;
    pqf_block_role(blockIdx.x, f.n_groups, slab, group);
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(f.table8 + (uint64_t)group * 256 * SLOTS);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = threadIdx.x; i < 256 * SLOTS / 4; i += PQF_THREADS) dst[i] = src[i];
    }
    // the routing operand: B[K][n] = 1 iff n == 4 (K / 16) + K % 4.  Lane n + 16 kg holds B[16 kg .. 16 kg + 15][n]: ones at bytes k, 4 + k, 8 + k, 12 + k
    // when n / 4 == kg (k = n % 4), zeros otherwise.
    const uint32_t bword = (((uint32_t)lane & 15u) >> 2) == ((uint32_t)lane >> 4) ? (1u << (8 * ((uint32_t)lane & 3u))) : 0u;
    const pqf_i32x4 B = {(int)bword, (int)bword, (int)bword, (int)bword};
    // D[r] of this lane = the sum, over the chunks so far, of query (lane & 3)'s entries for the row of lane dl0 + r
    const uint32_t dl0 = 4 * ((uint32_t)lane >> 4) + 16 * (((uint32_t)lane & 15u) >> 2);
    const uint32_t my_q = group * 4 + ((uint32_t)lane & 3u);
    const int bias = 128 * (int)M_PAD;                        // every slot of the table holds (entry - 128)
    const int my_thr = f.thr[my_q] - bias;
    __syncthreads();
    const uint32_t i = (uint32_t)lane & 31u;
    const uint32_t i4 = i * 4 + (uint32_t)(uintptr_t)(pqf_lds_byte *)smem;     // (the table starts at LDS address 0: 4 i stays below the row pitch)
    const uint32_t wave_global = blockIdx.x * PQF_WAVES + (uint32_t)wave;
    uint4 *my_list = f.wlist + (uint64_t)wave_global * f.wcap;
    uint32_t wcount = 0;
    const uint64_t n_wtiles = (a.n_cand + 63) / 64;
    for (uint64_t wt = (uint64_t)slab * PQF_WAVES + (uint32_t)wave; wt < n_wtiles; wt += (uint64_t)f.n_slabs * PQF_WAVES) {
        const uint64_t T = wt * 2 + ((uint32_t)lane >> 5);
        uint4 w[NPW];
#pragma unroll
        for (int p = 0; p < NPW; ++p) w[p] = f.rot[(T * NPW + p) * 32 + i];
        pqf_i32x4 acc = {0, 0, 0, 0};
        const uint32_t pitch = 1u << SH;
#pragma unroll
        for (int t4 = 0; t4 < (int)M_PAD / 4; ++t4) {
            pqf_i32x4 g;
            const uint4 &piece = w[t4 / 4];
            const uint32_t d = t4 % 4 == 0 ? piece.x : t4 % 4 == 1 ? piece.y : t4 % 4 == 2 ? piece.z : piece.w;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // byte e of d, times the row pitch, or'ed with the lane's slot offset (4 i < 128 <= pitch): two vector instructions per gather
                // (written out: the compiler's own choice for the two low bytes is shift + and + add)
                uint32_t addr;
                if (8 * e < (int)SH) {
                    const uint32_t shifted = d << (SH - 8 * e);
                    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(addr) : "v"(shifted), "s"(0xFFu << SH), "v"(i4));
                } else {
                    const uint32_t code = e == 3 ? d >> 24 : (d >> (8 * e)) & 0xFFu;
                    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(addr) : "v"(code), "s"(SH), "v"(i4));
                }
                // (an LDS address outright - the table's base is folded into i4 -, so that the slot offset becomes the instruction's immediate)
                g[e] = *reinterpret_cast<pqf_lds_int *>(addr + 4 * (4 * t4 + e));
            }
            acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(g, B, acc, 0, 0, 0);
        }
        if (__ballot(acc[0] >= my_thr || acc[1] >= my_thr || acc[2] >= my_thr || acc[3] >= my_thr)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint64_t id64 = wt * 64 + dl0 + (uint32_t)r;
                const bool h = id64 < a.n_cand && acc[r] >= my_thr;
                const uint64_t mk = __ballot(h);
                if (mk) {
                    const uint32_t pos = wcount + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull));
                    if (h && pos < f.wcap) my_list[pos] = make_uint4(~(uint32_t)id64, score_to_ord((float)(acc[r] + bias)), my_q, 0u);
                    wcount += (uint32_t)__popcll(mk);
                }
            }
        }
    }
    if (lane == 0) f.wcnt[wave_global] = wcount;
}

size_t pq_prefilter_table_bytes(uint32_t m, uint32_t nq) {
    const uint32_t m_pad = (m + 31) / 32 * 32;
    return (size_t)((nq + 3) / 4) * 256 * pqf_slots(m_pad) * 4;
}
uint32_t pq_prefilter_grid(int num_cus, uint32_t nq, uint32_t *n_slabs_out) {
    const uint32_t n_groups = (nq + 3) / 4;
    uint32_t n_slabs = std::max<uint32_t>(8, ((uint32_t)num_cus / n_groups) / 8 * 8);
    if (n_slabs_out) *n_slabs_out = n_slabs;
    return n_slabs * n_groups;
}
size_t pq_prefilter_wlists_counts_bytes(uint32_t grid) { return ((size_t)grid * PQF_WAVES * 4 + 255) / 256 * 256; }
size_t pq_prefilter_wlists_bytes(uint32_t grid, uint32_t wcap) { return pq_prefilter_wlists_counts_bytes(grid) + (size_t)grid * PQF_WAVES * wcap * 16; }

int32_t launch_pq_lut8(hipStream_t st, const void *d_luts, uint32_t q_stride, uint32_t nq, uint32_t m, uint32_t ncent, const uint64_t *d_gthr, void *d_table8,
                       int32_t *d_thr, float *d_band) {
    const uint32_t m_pad = (m + 31) / 32 * 32;
    QMX_REQUIRE(pq_prefilter_shape_ok(m, ncent), QMX_ERR_NOT_SUPPORTED, "PQ prefilter: m = %u chunks / %u centroids", m, ncent);
    QMX_HIP(hipMemsetAsync(d_table8, 0x80, pq_prefilter_table_bytes(m, nq), st));
    // the unused query slots of the last group never produce candidates
    const uint32_t padded = (nq + 3) / 4 * 4;
    if (padded != nq) QMX_HIP(hipMemsetAsync(d_thr + nq, 0x7F, (size_t)(padded - nq) * 4, st));
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(pq_lut8_kernel, dim3(nq), dim3(256), 0, st, (const unsigned char *)d_luts, q_stride, nq, m, ncent, m_pad, d_gthr, (uint8_t *)d_table8, d_thr,
                       d_band);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

int32_t launch_pq_prefilter(hipStream_t st, const ScanArgs &a, const void *d_rot, const void *d_table8, const int32_t *d_thr, uint32_t nq, int num_cus,
                            void *d_wlists, uint32_t wcap, uint32_t *grid_out) {
    const uint32_t m = a.pq_m, m_pad = (m + 31) / 32 * 32;
    PqfArgs f;
    f.rot = (const uint4 *)d_rot;
    f.table8 = (const uint32_t *)d_table8;
    f.thr = d_thr;
    f.n_groups = (nq + 3) / 4;
    const uint32_t grid = pq_prefilter_grid(num_cus, nq, &f.n_slabs);
    f.wcnt = (uint32_t *)d_wlists;
    f.wlist = (uint4 *)((unsigned char *)d_wlists + pq_prefilter_wlists_counts_bytes(grid));
    f.wcap = wcap;
    if (grid_out) *grid_out = grid;
    const size_t lds = (size_t)256 * pqf_slots(m_pad) * 4;
    auto k2 = pq_prefilter_kernel<2>;
    auto k4 = pq_prefilter_kernel<4>;
    auto k6 = pq_prefilter_kernel<6>;
    static thread_local DeviceOnce attr_once;
    if (attr_once.need()) {
        QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k4), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k6), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_once.mark();
    }
    ::qmx::clear_stale_error();
    if (m_pad == 32) {
        QMX_NOTE_KERNEL(k2);
        hipLaunchKernelGGL(k2, dim3(grid), dim3(PQF_THREADS), lds, st, a, f);
    } else if (m_pad == 64) {
        QMX_NOTE_KERNEL(k4);
        hipLaunchKernelGGL(k4, dim3(grid), dim3(PQF_THREADS), lds, st, a, f);
    } else {
        QMX_NOTE_KERNEL(k6);
        hipLaunchKernelGGL(k6, dim3(grid), dim3(PQF_THREADS), lds, st, a, f);
    }
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

}  // namespace qmx

// pq.hip — EncodedVectorsPQ (product quantization) on device.
//
// Reference (lib/quantization/src/encoded_vectors_pq.rs):
//   encode_query (LUT)   :519-541   LUT[c][j] = DistanceType::distance(query chunk c, centroid j chunk c), negated if invert
//                                    (DistanceType::distance: encoded_vectors.rs:119-127, sequential f32 sum, mul/add un-fused)
//   score_point_sse      :409-443   4 f32 lane accumulators (lane j takes chunks j, j+4, ...), (l0+l2)+(l1+l3), sequential tail
//   score_internal       :574-618   centroid <-> centroid distances summed over chunks
//   encode_vector        :301-329   per chunk L2 argmin over the centroids, first minimum wins
// Rows are `m` code bytes (one centroid index per chunk); the LUT of one query is m x n_centroids f32
// (96 KiB at m = 96) and lives in LDS for the scan.
//
// MFMA is used in exactly one place: the LUT build for Dot/Cosine (`lut_mfma`), a genuine dense
// contraction [nq x chunk] . [chunk x 256] per chunk with f32 inputs (v_mfma_f32_32x32x2_f32: an fmaf
// chain, so <= 1e-5 relative to the reference's mul+add chain; the exact-order VALU kernel is the
// bit-parity variant).
#include "hnsw_build.hpp"

namespace qmx {

struct PqGeom {
    uint32_t dim, chunk, m, ncent;
    int kind;     // 0 dot/cosine, 1 L1, 2 L2
    int invert;
};

__device__ __forceinline__ float pq_term(int kind, float a, float b) {
    if (kind == 0) return a * b;
    const float d = a - b;
    return kind == 1 ? __builtin_fabsf(d) : d * d;
}

// ------------------------------------------------------------------------------------------
// LUT build, reference order.  grid (m, nq), one thread per centroid.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pq_lut_kernel(PqGeom g, const float *queries /*[nq][dim] preprocessed*/,
                                                     const float *centroids /*[ncent][dim]*/, float *lut /*[nq][m][ncent]*/) {
    __shared__ float sub[256];
    const uint32_t c = blockIdx.x, q = blockIdx.y;
    const uint32_t lo = c * g.chunk, hi = min(lo + g.chunk, g.dim);
    if (threadIdx.x < hi - lo) sub[threadIdx.x] = queries[(uint64_t)q * g.dim + lo + threadIdx.x];   // chunk <= 256 (host-checked)
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < g.ncent; j += 256) {
        const float *cen = centroids + (uint64_t)j * g.dim + lo;
        float s = -0.0f;
        for (uint32_t i = 0; i < hi - lo; ++i) s += pq_term(g.kind, sub[i], cen[i]);
        lut[((uint64_t)q * g.m + c) * g.ncent + j] = g.invert ? -s : s;
    }
}

// ------------------------------------------------------------------------------------------
// LUT build on the matrix cores (Dot / Cosine only): per chunk c, C[q][j] = sum_i Q[q][lo+i] * Cen[j][lo+i].
// One wavefront computes a 32 (queries) x 32 (centroids) tile with v_mfma_f32_32x32x2_f32:
//   A (32 x 2): lane l supplies Q[q0 + l%32][lo + k0 + l/32]
//   B (2 x 32): lane l supplies Cen[j0 + l%32][lo + k0 + l/32]
//   C (32 x 32): lane l, register v holds row 8*(v/4) + 4*(l/32) + v%4, column l%32
// grid (m, ceil(nq/32)), block = 4 waves, wave w covers centroid tiles w, w+4, ...
// ------------------------------------------------------------------------------------------
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void pq_lut_mfma_kernel(PqGeom g, uint32_t nq, const float *queries, const float *centroids,
                                                          float *lut) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t c = blockIdx.x;
    const uint32_t q0 = blockIdx.y * 32;
    const uint32_t lo = c * g.chunk, hi = min(lo + g.chunk, g.dim);
    const uint32_t len = hi - lo;
    const uint32_t qrow = q0 + (lane & 31);
    const uint32_t khalf = lane >> 5;
    for (uint32_t j0 = wave * 32; j0 < g.ncent; j0 += 128) {
        const uint32_t jcol = j0 + (lane & 31);
        floatx16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        for (uint32_t k0 = 0; k0 < len; k0 += 2) {
            const uint32_t k = k0 + khalf;
            const float a = (qrow < nq && k < len) ? queries[(uint64_t)qrow * g.dim + lo + k] : 0.0f;
            const float b = (jcol < g.ncent && k < len) ? centroids[(uint64_t)jcol * g.dim + lo + k] : 0.0f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (jcol < g.ncent) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const uint32_t q = q0 + 8 * (v / 4) + 4 * khalf + (v % 4);
                if (q < nq) {
                    const float s = acc[v];
                    lut[((uint64_t)q * g.m + c) * g.ncent + jcol] = g.invert ? -s : s;
                }
            }
        }
    }
}

// The same contraction with both operands staged in LDS: a block of 4 waves takes 128 queries of one chunk (wave w: queries 32 w .. 32 w + 31 against every
// centroid tile), the chunk's centroids [ncent][len] and the block's query chunks [128][len] are copied into LDS once, with coalesced loads (pq_lut_mfma_kernel
// reads each operand element from global memory per MFMA: 32 cache lines per load instruction, eight times per tile), rows padded to an odd stride.  Same k order,
// same instruction, same accumulators: the LUT carries the bits of pq_lut_mfma_kernel.  The kernel is bound by the LUT it WRITES (nq x m x ncent x 4 B: 805 MB
// for 8192 queries of C4), not by the matrix cores (2 x ncent x dim flop per query: 6.4 GFLOP for the same batch = 0.04 ms of the f32 MFMA peak).
constexpr uint32_t PQ_LUT_QB = 128;
static inline size_t pq_lut_lds_bytes(const PqGeom &g) {
    const uint32_t len2 = (g.chunk + 1) & ~1u;
    return (size_t)(g.ncent + PQ_LUT_QB) * (len2 + 1) * sizeof(float);
}
__global__ __launch_bounds__(256) void pq_lut_mfma_lds_kernel(PqGeom g, uint32_t nq, const float *queries, const float *centroids, float *lut) {
    extern __shared__ float pq_lut_sm[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t c = blockIdx.x;
    const uint32_t q0 = blockIdx.y * PQ_LUT_QB;
    const uint32_t lo = c * g.chunk, hi = min(lo + g.chunk, g.dim);
    const uint32_t len = hi - lo;
    const uint32_t len2 = (g.chunk + 1) & ~1u, LP = len2 + 1;
    float *cs = pq_lut_sm, *qs = pq_lut_sm + (size_t)g.ncent * LP;
    for (uint32_t idx = threadIdx.x; idx < g.ncent * len2; idx += 256) {
        const uint32_t j = idx / len2, i = idx % len2;
        cs[j * LP + i] = i < len ? centroids[(uint64_t)j * g.dim + lo + i] : 0.0f;
    }
    for (uint32_t idx = threadIdx.x; idx < PQ_LUT_QB * len2; idx += 256) {
        const uint32_t r = idx / len2, i = idx % len2;
        qs[r * LP + i] = (q0 + r < nq && i < len) ? queries[(uint64_t)(q0 + r) * g.dim + lo + i] : 0.0f;
    }
    __syncthreads();
    const uint32_t qw = q0 + 32u * (uint32_t)wave;
    if (qw >= nq) return;
    const uint32_t khalf = lane >> 5;
    const float *qrow = qs + (32u * (uint32_t)wave + (uint32_t)(lane & 31)) * LP;
    for (uint32_t j0 = 0; j0 < g.ncent; j0 += 32) {
        const uint32_t jcol = j0 + (lane & 31);
        const float *crow = cs + (jcol < g.ncent ? jcol : 0) * LP;
        floatx16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        for (uint32_t k0 = 0; k0 < len; k0 += 2) {
            const float a = qrow[k0 + khalf];                       // (columns past len are zero in LDS, queries past nq too)
            const float b = jcol < g.ncent ? crow[k0 + khalf] : 0.0f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (jcol < g.ncent) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const uint32_t q = qw + 8 * (v / 4) + 4 * khalf + (v % 4);
                if (q < nq) {
                    const float s = acc[v];
                    lut[((uint64_t)q * g.m + c) * g.ncent + jcol] = g.invert ? -s : s;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// score of one code row against one LUT, in score_point_sse order
// ------------------------------------------------------------------------------------------
template <class LutPtr>
__device__ __forceinline__ float pq_score_row(const LutPtr lut, const uint8_t *codes, uint32_t m, uint32_t ncent) {
    float l0 = 0.0f, l1 = 0.0f, l2 = 0.0f, l3 = 0.0f;
    const uint32_t m4 = m & ~3u;
    uint32_t c = 0;
    if ((reinterpret_cast<uintptr_t>(codes) & 3) == 0) {
        for (; c < m4; c += 4) {
            const uint32_t w = *reinterpret_cast<const uint32_t *>(codes + c);
            l0 += lut[(c + 0) * ncent + (w & 0xFF)];
            l1 += lut[(c + 1) * ncent + ((w >> 8) & 0xFF)];
            l2 += lut[(c + 2) * ncent + ((w >> 16) & 0xFF)];
            l3 += lut[(c + 3) * ncent + (w >> 24)];
        }
    } else {
        for (; c < m4; c += 4) {
            l0 += lut[(c + 0) * ncent + codes[c]];
            l1 += lut[(c + 1) * ncent + codes[c + 1]];
            l2 += lut[(c + 2) * ncent + codes[c + 2]];
            l3 += lut[(c + 3) * ncent + codes[c + 3]];
        }
    }
    float sum = (l0 + l2) + (l1 + l3);        // sum64 = sum128 + movehl; sum32 = sum64[0] + sum64[1]
    for (; c < m; ++c) sum += lut[c * ncent + codes[c]];
    return sum;
}

// ------------------------------------------------------------------------------------------
// Brute-force scan over PQ codes.  blockIdx.y = query (its LUT staged in LDS when it fits),
// blockIdx.x = row slab; one lane per row.  Consecutive blocks differ in the query first, so the
// blocks resident at any moment share row slabs through L2 / Infinity Cache.
// ------------------------------------------------------------------------------------------
constexpr int PQ_BLOCK = 1024;

template <bool LDS_LUT, bool HAS_IDS, int MODE>
__global__ __launch_bounds__(PQ_BLOCK) void pq_scan_kernel(const ScanArgs a, uint32_t n_slabs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint64_t sh_keys[PQ_BLOCK / WAVE][WAVE];
    constexpr int NW = PQ_BLOCK / WAVE;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // block -> (slab, query): query varies fastest.  The exact pass behind the 6-bit prefilter (pq_prefilter.hip: run_if = the number of queries whose
    // lists overflowed, q_map = those queries, packed) is launched for the worst case - every query of the batch - and shares its grid among the queries
    // that are listed: one overflowing query of 128 is scanned by every block of the launch, not by its 1 / 128th (12 ms -> 0.1 ms at 10 M x 96)
    uint32_t nq_eff = a.nq, pstride = a.partial_qt;
    if (a.run_if) {
        const uint32_t cnt = (uint32_t)*a.run_if;
        if (cnt == 0) return;
        if (a.q_map) {
            nq_eff = cnt < a.nq ? cnt : a.nq;
            n_slabs = gridDim.x / nq_eff;
            pstride = nq_eff;
            if (blockIdx.x >= n_slabs * nq_eff) return;
        }
    }
    const uint32_t q = blockIdx.x % nq_eff;
    const uint32_t slab = blockIdx.x / nq_eff;
    const uint32_t m = a.pq_m, ncent = a.pq_ncent;
    const uint32_t qsrc = a.q_map ? a.q_map[q] : q;          // (packed exact pass: list q belongs to query q_map[q] of the batch)
    if (qsrc == 0xFFFFFFFFu) return;
    const float *glut = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(a.queries) + (uint64_t)qsrc * a.q_stride);
    if (LDS_LUT) {
        const uint4 *src = reinterpret_cast<const uint4 *>(glut);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        const uint32_t n16 = m * ncent / 4;
        for (uint32_t i = threadIdx.x; i < n16; i += PQ_BLOCK) dst[i] = src[i];
        for (uint32_t i = n16 * 4 + threadIdx.x; i < m * ncent; i += PQ_BLOCK) reinterpret_cast<float *>(smem)[i] = glut[i];
        __syncthreads();
    }
    const float *slut = reinterpret_cast<const float *>(smem);
    const uint8_t *rows = reinterpret_cast<const uint8_t *>(a.rows);
    const int top = (int)a.top;
    uint64_t list = 0;
    for (uint64_t base = ((uint64_t)slab * NW + wave) * WAVE; base < a.n_cand; base += (uint64_t)n_slabs * NW * WAVE) {
        const uint64_t cnd = base + lane;
        bool valid = cnd < a.n_cand;
        uint32_t id = HAS_IDS ? a.ids[valid ? cnd : 0] : (uint32_t)cnd;
        if (HAS_IDS && valid && id >= a.n_rows) {
            *a.err_flag = 1;
            valid = false;
        }
        if (!valid) id = 0;
        float score = 0.0f;
        if (a.n_rows) {
            const uint8_t *codes = rows + (uint64_t)id * a.row_stride;
            score = LDS_LUT ? pq_score_row(slut, codes, m, ncent) : pq_score_row(glut, codes, m, ncent);
        }
        if (MODE == SCAN_SCORES) {
            if (valid) a.scores[(uint64_t)q * a.scores_stride + cnd] = score;
        } else {
            const uint64_t key = make_key(score, id);
            bool c = valid && key > readlane_u64(list, top - 1);
            if (__ballot(c)) {
                c = c && a.del.live(id) && (!a.key_bound || key < a.key_bound[q]);
                uint64_t mask = __ballot(c);
                while (mask) {
                    const int src = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    const uint64_t nk = readlane_u64(key, src);
                    if (nk > readlane_u64(list, top - 1)) wave_list_insert(list, nk, lane);
                }
            }
        }
    }
    if (MODE == SCAN_SCORES) return;
    sh_keys[wave][lane] = list;
    __syncthreads();
    if (wave == 0) {
        uint64_t merged = sh_keys[0][lane];
        for (int w = 1; w < NW; ++w) {
            const uint64_t key = sh_keys[w][lane];
            uint64_t mask = __ballot(key > readlane_u64(merged, top - 1));
            while (mask) {
                const int src = __builtin_ctzll(mask);
                mask &= mask - 1;
                const uint64_t nk = readlane_u64(key, src);
                if (nk > readlane_u64(merged, top - 1)) wave_list_insert(merged, nk, lane);
            }
        }
        if (lane < top) a.partial[((uint64_t)slab * pstride + q) * top + lane] = merged;
    }
}

template <bool LDS_LUT, bool HAS_IDS, int MODE>
static int32_t launch_pq_scan_inst(hipStream_t st, const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    const size_t lds = LDS_LUT ? (((size_t)a.pq_m * a.pq_ncent * 4 + 15) & ~(size_t)15) : 0;
    auto kfn = pq_scan_kernel<LDS_LUT, HAS_IDS, MODE>;
    static thread_local DeviceOnce attr_once;
    if (LDS_LUT && attr_once.need()) {
        QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));   // + 8 KiB static
        attr_once.mark();
    }
    // slabs: enough blocks to fill the chip with nq queries each, bounded by the work
    uint64_t want = (a.n_cand + PQ_BLOCK - 1) / PQ_BLOCK;
    uint64_t cap = std::max<uint64_t>(1, ((uint64_t)num_cus * 2 + a.nq - 1) / a.nq);
    uint32_t slabs = (uint32_t)std::max<uint64_t>(1, std::min(want, cap));
    if (grid_out) {
        if (*grid_out && MODE == SCAN_TOPK && slabs > *grid_out) slabs = *grid_out;
        *grid_out = slabs;
    }
    ::qmx::clear_stale_error();
    QMX_NOTE_KERNEL(kfn);
    hipLaunchKernelGGL(kfn, dim3(slabs * a.nq), dim3(PQ_BLOCK), lds, st, a, slabs);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

int32_t launch_scan_pq(hipStream_t st, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    const bool lds = (size_t)a.pq_m * a.pq_ncent * 4 <= 150 * 1024;
    const bool ids = a.ids != nullptr;
#define QMX_PQ_CASE(L, I, M) \
    if (lds == L && ids == I && mode == M) return launch_pq_scan_inst<L, I, M>(st, a, num_cus, grid_out);
    QMX_PQ_CASE(true, false, SCAN_TOPK)
    QMX_PQ_CASE(true, true, SCAN_TOPK)
    QMX_PQ_CASE(true, false, SCAN_SCORES)
    QMX_PQ_CASE(true, true, SCAN_SCORES)
    QMX_PQ_CASE(false, false, SCAN_TOPK)
    QMX_PQ_CASE(false, true, SCAN_TOPK)
    QMX_PQ_CASE(false, false, SCAN_SCORES)
    QMX_PQ_CASE(false, true, SCAN_SCORES)
#undef QMX_PQ_CASE
    return QMX_ERR_OTHER;
}

// pair scoring (HNSW hops / ragged lists): one lane per item, LUT read through L2
__global__ __launch_bounds__(256) void pq_pair_kernel(const ScanArgs a, const PairSel sel, uint64_t n_items) {
    const uint8_t *rows = reinterpret_cast<const uint8_t *>(a.rows);
    for (uint64_t item = (uint64_t)blockIdx.x * 256 + threadIdx.x; item < n_items; item += (uint64_t)gridDim.x * 256) {
        const uint32_t qi = sel.query_of(item);
        if (!sel.live(item, qi)) continue;
        const uint32_t id = a.ids[item];
        if (id >= a.n_rows || qi >= a.nq) {
            *a.err_flag = 1;
            continue;
        }
        const float *lut = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(a.queries) + (uint64_t)qi * a.q_stride);
        a.scores[item] = pq_score_row(lut, rows + (uint64_t)id * a.row_stride, a.pq_m, a.pq_ncent);
    }
}
int32_t launch_pairs_pq(hipStream_t st, const ScanArgs &a, const PairSel &sel, uint64_t n_items, int num_cus) {
    if (n_items == 0) return QMX_OK;
    uint64_t want = (n_items + 255) / 256;
    uint32_t grid = (uint32_t)std::min<uint64_t>(want, (uint64_t)num_cus * 8);
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(pq_pair_kernel, dim3(grid), dim3(256), 0, st, a, sel, n_items);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

// hop_ids[0..k) -> the ids whose 8-bit upper bound reaches the score of `bound` (the key of the beam's worst entry; 0: the beam is not full, everything
// passes), in order, at hop_ids[0..k'); returns k'.  See HopPQ below for what it is for.  pq8 = [L, Es, step: f64][usable: u32][pad][m x 256 bytes] (LDS).
constexpr double PQ_WALK_ROUND = 0.5 + 1.0e-4;           // = PQF_ROUND of pq_prefilter.hip: rint's half + the f32 evaluation of the quotient
static __device__ __forceinline__ uint32_t pq_hop_prefilter(const ScanArgs &a, const unsigned char *pq8, uint32_t *hop_ids, uint32_t k, uint64_t bound, int lane) {
    const double L = *reinterpret_cast<const double *>(pq8), Es = *reinterpret_cast<const double *>(pq8 + 8), step = *reinterpret_cast<const double *>(pq8 + 16);
    const uint32_t usable = *reinterpret_cast<const uint32_t *>(pq8 + 24);
    if (!usable || bound == 0) return k;
    const uint32_t m = a.pq_m;
    const double t = __builtin_floor(((double)key_score(bound) - L - Es) / step - PQ_WALK_ROUND * (double)m) - 1.0;
    if (!(t > 0.0)) return k;                                   // (also NaN bounds: nothing is dropped)
    const uint32_t a_min = t > 1.0e9 ? 1000000000u : (uint32_t)t;
    const unsigned char *tab = pq8 + 32;
    const int sub = lane & 3, g = lane >> 2;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint32_t kept = 0;
    __syncthreads();      // hop_ids written by other lanes
    for (uint32_t base = 0; base < k; base += 16) {
        const uint32_t j = base + (uint32_t)g;
        const bool on = j < k;
        const uint32_t id = hop_ids[on ? j : 0];
        const uint8_t *codes = reinterpret_cast<const uint8_t *>(a.rows) + (uint64_t)id * a.row_stride;
        uint32_t sum = 0;
        for (uint32_t c = (uint32_t)sub; c < m; c += 4) sum += tab[c * 256u + codes[c]];
        sum += (uint32_t)__shfl_xor((int)sum, 1, 64);
        sum += (uint32_t)__shfl_xor((int)sum, 2, 64);
        const bool pass = on && sub == 0 && sum >= a_min;
        const uint64_t pm = __ballot(pass);
        if (pass) hop_ids[kept + (uint32_t)__popcll(pm & lt_mask)] = id;      // (positions <= j: never an entry a later pass still has to read)
        kept += (uint32_t)__popcll(pm);
    }
    return kept;
}

// HNSW hop scorer: 4 lanes per code row, lane `sub` owns SSE lane `sub` of score_point_sse (chunks
// sub, sub+4, ... added in order), the quad is folded as (l0 + l2) + (l1 + l3) like pq_score_row.
struct HopPQ {
    static constexpr int LPI = 4;
    static constexpr bool MULTI = false;
    static __device__ __forceinline__ float score(const ScanArgs &a, const unsigned char *qp, uint32_t id, int sub) {
        const float *lut = reinterpret_cast<const float *>(qp);
        const uint8_t *codes = reinterpret_cast<const uint8_t *>(a.rows) + (uint64_t)id * a.row_stride;
        const uint32_t m = a.pq_m, ncent = a.pq_ncent, m4 = m & ~3u;
        float l = 0.0f;
        uint32_t c = 0;
        if ((reinterpret_cast<uintptr_t>(codes) & 15) == 0 && m4 <= 128) {
            // the walk is latency-bound (one search per CU when the LUT fills the LDS): fetch the whole code row with
            // independent loads first (one HBM round trip), then 8 LUT gathers in flight per 8 adds, adds in chunk order
            uint4 w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) w[k] = (uint32_t)(16 * k) < m4 ? *reinterpret_cast<const uint4 *>(codes + 16 * k) : make_uint4(0, 0, 0, 0);
            // ... then ALL the lane's LUT gathers (m / 4 <= 32) before the first add: entries past m4 read entry 0 of the table (a valid address, the
            // value unused), so the gathers are straight-line code and one round trip to L2 instead of one per block of eight
            float v[4][8];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const uint32_t ws[8] = {w[2 * h].x, w[2 * h].y, w[2 * h].z, w[2 * h].w, w[2 * h + 1].x, w[2 * h + 1].y, w[2 * h + 1].z, w[2 * h + 1].w};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t cc = 32 * h + 4 * k;
                    const uint32_t at = cc < m4 ? (cc + (uint32_t)sub) * ncent + ((ws[k] >> (8 * sub)) & 0xFF) : 0u;
                    v[h][k] = lut[at];
                }
            }
#pragma unroll
            for (int h = 0; h < 4; ++h)
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if ((uint32_t)(32 * h + 4 * k) < m4) l += v[h][k];
            c = m4;
        } else if ((reinterpret_cast<uintptr_t>(codes) & 3) == 0) {
            for (; c < m4; c += 4) {
                const uint32_t w = *reinterpret_cast<const uint32_t *>(codes + c);
                l += lut[(c + (uint32_t)sub) * ncent + ((w >> (8 * sub)) & 0xFF)];
            }
        } else {
            for (; c < m4; c += 4) l += lut[(c + (uint32_t)sub) * ncent + codes[c + (uint32_t)sub]];
        }
        const float x = l + dpp_f32<DPP_QUAD_XOR2>(l);          // lane 0: l0 + l2, lane 1: l1 + l3
        float sum = x + dpp_f32<DPP_QUAD_XOR1>(x);              // lane 0: (l0 + l2) + (l1 + l3)
        for (c = m4; c < m; ++c) sum += lut[c * ncent + codes[c]];
        return sum;
    }
    // ---- the hop prefilter (round 5) ----
    // A search scores ~3 600 candidates and the beam admits a fraction of them; every exact score is m gathers from the search's own 96 KiB LUT through L2
    // (61 x the useful bytes in HBM traffic).  The LUT quantised to 8 bits (pq_walk_lut8_kernel: q_cj = rint((LUT[c][j] - lo_c) / step), one step per query)
    // fits the LDS, and a candidate's integer sum A bounds its exact score from above: S <= L + step (A + PQF_ROUND m) + Es (the bound of pq_prefilter.hip).
    // A candidate whose upper bound is below the score of the beam's worst entry cannot be inserted - the reference's process_candidate would reject it on
    // its exact score - so it is dropped WITHOUT the exact score; the others are compacted in link order and scored exactly as before.  The walk is the same
    // walk (same inserts in the same order, same counters: the dropped candidates still count as scored, as in the reference where they were).
    static constexpr bool HOP_PREFILTER = true;
    static __device__ __forceinline__ uint32_t prefilter(const ScanArgs &a, const unsigned char *pq8, uint32_t *hop_ids, uint32_t k, uint64_t bound, int lane) {
        return pq_hop_prefilter(a, pq8, hop_ids, k, bound, lane);
    }
};
int32_t launch_hnsw_pq(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu) {
    return launch_hnsw_hop<HopPQ>(st, a, h, grid, per_cu);
}

// One block per query, thread j = centroid j: the 8-bit image of the query's f32 LUT [m][ncent] for HopPQ::prefilter.
//   lo_c = min_j LUT[c][j], R = max_c (max_j - min_j), step = R / 255, q_cj = rint((LUT[c][j] - lo_c) / step) in 0..255 (centroids past ncent: 255, never read)
//   header: L = sum_c lo_c, Es = (m + 1) 2^-24 sum_c max_j |LUT[c][j]| (the f32 rounding of the exact sum), step; usable = 0 for flat or non-finite tables
// (the body: `lut` may point at global memory or at LDS; 256 threads)
static __device__ __forceinline__ void pq_walk_lut8_body(const float *lut, uint32_t m, uint32_t ncent, unsigned char *dst, float *sh_lo, float *sh_hi, float *sh_ab, int *sh_bad) {
    const uint32_t j = threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (j == 0) *sh_bad = 0;
    __syncthreads();
    int bad = 0;
    for (uint32_t c = (uint32_t)wave; c < m; c += 4) {
        float mn = __builtin_inff(), mx = -__builtin_inff(), ab = 0.0f;
        for (uint32_t jj = (uint32_t)lane; jj < ncent; jj += 64) {
            const float v = lut[(uint64_t)c * ncent + jj];
            if ((v != v) || !(__builtin_fabsf(v) < 3.0e38f)) bad = 1;
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
    if (bad) *sh_bad = 1;
    __syncthreads();
    float R = 0.0f, E = 0.0f;
    double L = 0.0;
    for (uint32_t c = 0; c < m; ++c) {          // (every thread: the sums in chunk order)
        R = __builtin_fmaxf(R, sh_hi[c] - sh_lo[c]);
        E += sh_ab[c];
        L += (double)sh_lo[c];
    }
    const bool flat = !(R > 0.0f) || !(R < 3.0e38f) || *sh_bad;
    const float inv_step = flat ? 0.0f : 255.0f / R;
    for (uint32_t c0 = 0; c0 < m; c0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (c0 + u < m && j < ncent) ? lut[(uint64_t)(c0 + u) * ncent + j] : 0.0f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t c = c0 + (uint32_t)u;
            if (c < m) {
                float x = __builtin_rintf((v[u] - sh_lo[c]) * inv_step);
                x = __builtin_fminf(__builtin_fmaxf(x, 0.0f), 255.0f);
                dst[32u + c * 256u + j] = (j < ncent && !flat) ? (unsigned char)(uint32_t)x : (unsigned char)255;
            }
        }
    }
    if (j == 0) {
        double *hd = reinterpret_cast<double *>(dst);
        hd[0] = L;
        hd[1] = (double)(m + 1) * 5.9604644775390625e-08 * (double)E;
        hd[2] = flat ? 1.0 : (double)R / 255.0;
        reinterpret_cast<uint32_t *>(dst)[6] = flat ? 0u : 1u;
        reinterpret_cast<uint32_t *>(dst)[7] = 0u;
    }
}
__global__ __launch_bounds__(256) void pq_walk_lut8_kernel(const unsigned char *luts, uint32_t q_stride, uint32_t m, uint32_t ncent, unsigned char *out, uint32_t out_stride) {
    __shared__ float sh_lo[128], sh_hi[128], sh_ab[128];
    __shared__ int sh_bad;
    const uint32_t q = blockIdx.x;
    pq_walk_lut8_body(reinterpret_cast<const float *>(luts + (uint64_t)q * q_stride), m, ncent, out + (uint64_t)q * out_stride, sh_lo, sh_hi, sh_ab, &sh_bad);
}
int32_t launch_pq_walk_lut8(hipStream_t st, const void *d_luts, uint32_t q_stride, uint32_t nq, uint32_t m, uint32_t ncent, void *d_out) {
    if (nq == 0) return QMX_OK;
    QMX_REQUIRE(m >= 1 && m <= 128 && ncent >= 1 && ncent <= 256, QMX_ERR_NOT_SUPPORTED, "pq walk prefilter: m %u, %u centroids", m, ncent);
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(pq_walk_lut8_kernel, dim3(nq), dim3(256), 0, st, (const unsigned char *)d_luts, q_stride, m, ncent, (unsigned char *)d_out, pq_walk_lut8_stride(m));
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

// ------------------------------------------------------------------------------------------
// The same hop scorer WITHOUT the LUT (round 4).  A walk with HopPQ gathers 4-byte entries of its search's own LUT (96 KiB at m = 96): ~4 000 concurrent
// searches keep 384 MB of LUTs in flight, every gather pulls a 64-byte sector through the fabric for 4 useful bytes (PMC: 175 GB moved for 2.9 GB of
// codes, 0.015 of HBM on useful bytes, three rounds running) - and the LUTs themselves (768 MB per 8 192 searches) are written and read back.
// A LUT entry is lut[c][k] = sum_i term(q[lo + i], centroid[k][lo + i]) (pq_lut_kernel: from -0.0, one multiply and one add per coordinate, in order;
// encoded_vectors_pq.rs:519-541).  Here the lane that needs entry (c, code) computes exactly that chain from the codebook: 64 contiguous bytes of a
// 1.5 MB table that every search shares and that stays in L2, against the query's chunk in LDS (the entry of a search is its 6 KiB preprocessed vector,
// not a LUT).  Same bits as the exact LUT (a caller who allowed the matrix-core LUT accepted 1e-5 of it), same order of the m adds (lane `sub` of a quad
// owns chunks sub, sub + 4, ...: score_point_sse's lanes).  CHUNK floats per chunk, dim = m x CHUNK exactly; rounds of R chunk steps keep
// R x CHUNK / 4 sixteen-byte loads in flight per lane.
// ------------------------------------------------------------------------------------------
typedef float f32x4s __attribute__((ext_vector_type(4)));
template <int CHUNK, int SPLIT, int R>      // SPLIT lanes per SSE lane (4 SPLIT lanes per code row), R chunk steps per round
struct HopPQDirect {
    // A hop brings 5 - 10 fresh candidates: with four lanes per row most of the wave idles and each lane walks 24 chunk steps - at R steps per round trip
    // (R x CHUNK / 4 sixteen-byte registers of codebook in flight) a hop is a dozen dependent trips to L2.  So an SSE lane's steps are dealt to SPLIT
    // lanes: part p computes the entries of steps [p spp, (p + 1) spp) - the expensive part, in parallel - then the running sum of the SSE lane is handed
    // from part to part and every part adds its entries in step order: the same chain of adds, one lane at a time.
    static constexpr int LPI = 4 * SPLIT;
    static constexpr bool MULTI = false;
    static constexpr bool INTERNAL_QOFF = false;
    static constexpr bool INTERNAL_NORM = false;
    // The hop prefilter of HopPQ serves this walk too (round 6): the 8-bit image of the search's EXACT-ORDER LUT (HnswArgs::pq8, staged in LDS) bounds the
    // scores recomputed here from the codebook - the same entries, pq_lut_kernel's chain -, so a candidate whose bound stays below the beam's worst score
    // leaves without the codebook arithmetic, and the survivors' entries come from the 1.5 MB codebook every search shares in L2 instead of from a
    // 96 KiB per-search LUT that misses it: the walk's memory requests are its code rows and link rows.
    static constexpr bool HOP_PREFILTER = true;
    static __device__ __forceinline__ uint32_t prefilter(const ScanArgs &a, const unsigned char *pq8, uint32_t *hop_ids, uint32_t k, uint64_t bound, int lane) {
        return pq_hop_prefilter(a, pq8, hop_ids, k, bound, lane);
    }
    static constexpr int V = CHUNK / 4;                      // 16-byte pieces of a chunk
    static constexpr int SPP = 32 / SPLIT;                   // steps per part at most (m <= 128)
    template <int KIND>
    static __device__ __forceinline__ void entries(const ScanArgs &a, const float *q, const uint8_t *codes, uint32_t p0, uint32_t n_mine, int quad_lane, float (&t)[SPP]) {
        const float *cent = a.pq_centroids;
        const uint32_t dim = a.pq_dim;
        uint32_t code[SPP];
#pragma unroll
        for (int r = 0; r < SPP; ++r) code[r] = (uint32_t)r < n_mine ? codes[4u * (p0 + (uint32_t)r) + (uint32_t)quad_lane] : 0u;      // one round trip: the lane's code bytes
#pragma unroll
        for (int r0 = 0; r0 < SPP; r0 += R) {
            if (__ballot((uint32_t)r0 < n_mine) == 0) break;
            f32x4s cv[R][V];
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const int r = r0 + rr;
                // (past the lane's steps: chunk 0 of centroid 0 - a valid address, the value unused - so the loads are straight-line code)
                const bool on = r < SPP && (uint32_t)r < n_mine;
                const uint32_t c = on ? 4u * (p0 + (uint32_t)r) + (uint32_t)quad_lane : 0u;
                const float *p = cent + (size_t)(on ? code[r < SPP ? r : 0] : 0u) * dim + (size_t)c * CHUNK;
#pragma unroll
                for (int v = 0; v < V; ++v) cv[rr][v] = *reinterpret_cast<const f32x4s *>(p + 4 * v);
            }
            float sr[R];
#pragma unroll
            for (int rr = 0; rr < R; ++rr) sr[rr] = -0.0f;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                f32x4s qv[R];
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    const int r = r0 + rr;
                    const uint32_t c = (r < SPP && (uint32_t)r < n_mine) ? 4u * (p0 + (uint32_t)r) + (uint32_t)quad_lane : 0u;
                    qv[rr] = *reinterpret_cast<const f32x4s *>(q + (size_t)c * CHUNK + 4 * v);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int rr = 0; rr < R; ++rr) sr[rr] += pq_term(KIND, qv[rr][e], cv[rr][v][e]);
            }
#pragma unroll
            for (int rr = 0; rr < R; ++rr)
                if (r0 + rr < SPP) t[r0 + rr] = a.pq_invert ? -sr[rr] : sr[rr];
        }
    }
    // entry (c, code) alone (the chunks past the last whole quad of a row)
    static __device__ __forceinline__ float entry(const ScanArgs &a, const float *q, uint32_t c, uint32_t code) {
        const float *p = a.pq_centroids + (size_t)code * a.pq_dim + (size_t)c * CHUNK, *qc = q + (size_t)c * CHUNK;
        float s = -0.0f;
        for (int i = 0; i < CHUNK; ++i) s += pq_term((int)a.pq_kind, qc[i], p[i]);
        return a.pq_invert ? -s : s;
    }
    static __device__ __forceinline__ float score(const ScanArgs &a, const unsigned char *qp, uint32_t id, int sub) {
        const float *q = reinterpret_cast<const float *>(qp);
        const uint8_t *codes = reinterpret_cast<const uint8_t *>(a.rows) + (uint64_t)id * a.row_stride;
        const uint32_t m = a.pq_m, nsteps = m / 4;
        const int quad_lane = sub & 3, part = sub >> 2;
        const uint32_t spp = (nsteps + SPLIT - 1) / SPLIT;      // steps per part (the last part may hold fewer, or none)
        const uint32_t p0 = (uint32_t)part * spp;
        const uint32_t n_mine = p0 < nsteps ? (nsteps - p0 < spp ? nsteps - p0 : spp) : 0u;
        float t[SPP];
#pragma unroll
        for (int r = 0; r < SPP; ++r) t[r] = 0.0f;
        if (a.pq_kind == 0) entries<0>(a, q, codes, p0, n_mine, quad_lane, t);
        else if (a.pq_kind == 1) entries<1>(a, q, codes, p0, n_mine, quad_lane, t);
        else entries<2>(a, q, codes, p0, n_mine, quad_lane, t);
        // the SSE lane's sum, handed from part to part: l = ((0 + t_0) + t_1) + ... in step order
        float l = 0.0f;
#pragma unroll
        for (int p = 0; p < SPLIT; ++p) {
            if (p > 0) l = __shfl_up(l, 4, 64);      // (part p - 1 of the same row sits four lanes below)
            if (part == p) {
#pragma unroll
                for (int r = 0; r < SPP; ++r)
                    if ((uint32_t)r < n_mine) l += t[r];
            }
        }
        // the last part's quad holds the four SSE lanes' sums
        const float x = l + dpp_f32<DPP_QUAD_XOR2>(l);          // lane 0: l0 + l2, lane 1: l1 + l3
        float sum = x + dpp_f32<DPP_QUAD_XOR1>(x);              // lane 0: (l0 + l2) + (l1 + l3)
        for (uint32_t c = nsteps * 4; c < m; ++c) sum += entry(a, q, c, codes[c]);
        if (SPLIT > 1) sum = __shfl_down(sum, 4 * (SPLIT - 1), 64);      // to the row's first lane (the one whose score is stored)
        return sum;
    }
};
bool pq_direct_walk_ok(uint32_t dim, uint32_t m, uint32_t chunk, uint32_t ncent) {
    return (chunk == 16 || chunk == 8 || chunk == 4) && (uint64_t)m * chunk == dim && m <= 128 && ncent <= 256 && (uint64_t)dim * 4 <= 64 * 1024;
}
int32_t launch_hnsw_pq_direct(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu) {
    QMX_REQUIRE(a.pq_centroids && pq_direct_walk_ok(a.pq_dim, a.pq_m, a.pq_chunk, a.pq_ncent), QMX_ERR_BAD_ARG, "the LUT-free PQ walk does not take this codebook");
    // four lanes per SSE lane, two chunk steps (eight 16-byte loads) per round: the best of the shapes measured (profiles/r4_pq_direct_walk.md)
    // (behind the hop prefilter - ~4 survivors per hop instead of ~14 candidates - eight parts per SSE lane, one round trip per pass, measured SLOWER: 17.5
    // against 16.5 ms at 10 M x 1536 points, profiles/r6_walk_variants_ref_order.jsonl)
    if (a.pq_chunk == 16) return launch_hnsw_hop<HopPQDirect<16, 4, 2>>(st, a, h, grid, per_cu);
    if (a.pq_chunk == 8) return launch_hnsw_hop<HopPQDirect<8, 4, 4>>(st, a, h, grid, per_cu);
    return launch_hnsw_hop<HopPQDirect<4, 4, 8>>(st, a, h, grid, per_cu);
}

// multi-vector points over PQ inner rows (QuantizedMultivectorStorage<EncodedVectorsPQ>): MaxSim over the LUTs of the query's inner vectors, staged in LDS
int32_t launch_hnsw_maxsim_pq(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu) {
    return launch_hnsw_hop<HopMaxSim<HopPQ>>(st, a, h, grid, per_cu);
}
// ... with a custom query as the scorer: every example's LUT stays in global memory (read through L2, like the plain PQ walk's large LUTs)
int32_t launch_hnsw_custom_pq(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu) {
    return launch_hnsw_hop<HopCustom<HopPQ>>(st, a, h, grid, per_cu);
}

// ------------------------------------------------------------------------------------------
// HNSW build over a PQ segment (hnsw/build.rs:334-341 + point_scorer.rs:183-218).  EncodedVectorsPQ cannot turn a stored row
// into a query (encode_internal_vector -> None), so the searches of an insertion score through the LUT of the point's ORIGINAL
// vector (HopPQ over the batch's LUTs, made by api_hnsw.hip before phase 1) while everything stored <-> stored — the heuristic, the
// back links, an entry point at or below the new point's level — is EncodedVectorsPQ::score_internal (:574-618): the sum over
// chunks of the distance between the two rows' centroids.  Those chunk distances are tabulated once per segment
// (pair[c][i][j], m x 256 x 256 f32 = 25 MB at m = 96) with the reference's own inner loop, so a pair score is m table gathers
// added in chunk order: the bits of the reference.  One lane per stored row, the "query" row's codes are wave-uniform.
// ------------------------------------------------------------------------------------------
struct HopPQInternal {
    static constexpr int LPI = 1;
    static constexpr bool MULTI = false;
    static constexpr bool INTERNAL_QOFF = false;
    static constexpr bool INTERNAL_NORM = false;
    static constexpr bool ASYMMETRIC = true;
    static __device__ __forceinline__ float score(const ScanArgs &a, const unsigned char *qp, uint32_t id, int) {
        const uint8_t *cb = reinterpret_cast<const uint8_t *>(a.rows) + (uint64_t)id * a.row_stride;
        const uint32_t m = a.pq_m, nc = a.pq_ncent;
        float s = -0.0f;
        // sixteen table gathers in flight per lane, then their adds in chunk order (the reference's sum, bit for bit): one gather at a time - what a plain
        // loop over a runtime m compiles to - made a pair score a chain of m round trips to L2 / the Infinity Cache (the 25 MB table of m = 96 fits neither
        // a CU's L1 nor an XCD's L2) and the build's link phase the slowest kernel of C4 (round 3: 154 s per 10 M points)
        for (uint32_t c0 = 0; c0 < m; c0 += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const uint32_t c = c0 + (uint32_t)u < m ? c0 + (uint32_t)u : m - 1;      // (past the row: the last chunk again - a valid address, the value unused)
                v[u] = a.pq_pair[((uint64_t)c * nc + qp[c]) * nc + cb[c]];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (c0 + (uint32_t)u < m) s += v[u];
        }
        return a.pq_invert ? -s : s;
    }
};
struct HopPQBuild : HopPQ {
    static constexpr bool INTERNAL_QOFF = false;
    static constexpr bool INTERNAL_NORM = false;
};
// ------------------------------------------------------------------------------------------
// The same stored <-> stored score WITHOUT the pair table (round 4; with HopPQDirect for the insertion searches: a PQ build that gathers from no
// per-query or per-segment table at all; the default since it measured faster: 25.1 s against 32.0 s for 2 M x 1536 points, option hnsw_pq_table_build for the
// other).  PMC of a 2 M-point build (profiles/r3_pmc_traffic.md): the link phase moves 594 GB per launch and the search
// phase 347 GB - 100 TB per build, i.e. the build runs at the fabric's speed on 64-byte sectors of which it uses 4 bytes.  pair[c][i][j] is
// sum_k term(centroid[i][16 c + k], centroid[j][16 c + k]) from -0.0 (pq_pair_table_kernel): a LUT entry whose "query" is the other row's own
// centroid.  So the row that plays the query is DECODED once per hop into LDS (m chunks of its centroids: 6 KiB at d = 1536), every candidate's chunk
// entries are recomputed against it from the 1.5 MB codebook (16 lanes per candidate, six consecutive chunks each), and the single chain of m adds
// (`s = -0.0; s += entry[c]` in chunk order, then `invert`) is handed from lane to lane.  The policy scores a whole hop itself (hnsw.hpp hop_score).
// ------------------------------------------------------------------------------------------
template <int CHUNK>
struct HopPQInternalDirect {
    static constexpr int LPI = 16;
    static constexpr bool MULTI = false;
    static constexpr bool INTERNAL_QOFF = false;
    static constexpr bool INTERNAL_NORM = false;
    static constexpr bool ASYMMETRIC = true;
    static constexpr bool TQL1 = true;                     // (a policy that scores the hop as a wave: H::hop)
    static constexpr int V = CHUNK / 4;
    static constexpr int CPL = 8;                          // chunks per lane at most (m <= 128)
    static constexpr int R = 8 / V > 0 ? 8 / V : 1;        // chunks per round: eight 16-byte loads in flight per lane
    static __device__ __forceinline__ float score(const ScanArgs &, const unsigned char *, uint32_t, int) { return 0.0f; }   // (never called: hop() scores)
    template <int KIND>
    static __device__ __forceinline__ void entries(const ScanArgs &a, const float *q, const uint8_t *codes, uint32_t c0, uint32_t n_mine, float (&t)[CPL]) {
        const float *cent = a.pq_centroids;
        const uint32_t dim = a.pq_dim;
        uint32_t code[CPL];
#pragma unroll
        for (int r = 0; r < CPL; ++r) code[r] = (uint32_t)r < n_mine ? codes[c0 + (uint32_t)r] : 0u;
#pragma unroll
        for (int r0 = 0; r0 < CPL; r0 += R) {
            if (__ballot((uint32_t)r0 < n_mine) == 0) break;
            f32x4s cv[R][V];
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const int r = r0 + rr;
                const bool on = r < CPL && (uint32_t)r < n_mine;
                const uint32_t c = on ? c0 + (uint32_t)r : 0u;
                const float *p = cent + (size_t)(on ? code[r < CPL ? r : 0] : 0u) * dim + (size_t)c * CHUNK;
#pragma unroll
                for (int v = 0; v < V; ++v) cv[rr][v] = *reinterpret_cast<const f32x4s *>(p + 4 * v);
            }
            float sr[R];
#pragma unroll
            for (int rr = 0; rr < R; ++rr) sr[rr] = -0.0f;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                f32x4s qv[R];
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    const int r = r0 + rr;
                    const uint32_t c = (r < CPL && (uint32_t)r < n_mine) ? c0 + (uint32_t)r : 0u;
                    qv[rr] = *reinterpret_cast<const f32x4s *>(q + (size_t)c * CHUNK + 4 * v);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int rr = 0; rr < R; ++rr) sr[rr] += pq_term(KIND, qv[rr][e], cv[rr][v][e]);
            }
#pragma unroll
            for (int rr = 0; rr < R; ++rr)
                if (r0 + rr < CPL) t[r0 + rr] = sr[rr];
        }
    }
    static __device__ __forceinline__ void hop(const ScanArgs &a, const unsigned char *qp, const uint32_t *hop_ids, float *hop_scores, uint32_t k, int lane) {
        __shared__ __attribute__((aligned(16))) float qdec[128 * CHUNK];
        const uint32_t m = a.pq_m, dim = a.pq_dim;
        // the query row, decoded: chunk c of centroid qp[c]
        for (uint32_t i = (uint32_t)lane; i < m * V; i += 64) {
            const uint32_t c = i / V, v = i % V;
            *reinterpret_cast<f32x4s *>(qdec + (size_t)c * CHUNK + 4 * v) =
                *reinterpret_cast<const f32x4s *>(a.pq_centroids + (size_t)qp[c] * dim + (size_t)c * CHUNK + 4 * v);
        }
        __syncthreads();
        const int sub = lane & 15, g = lane >> 4;
        const uint32_t cpl = (m + 15) / 16;                       // consecutive chunks per lane
        const uint32_t c0 = (uint32_t)sub * cpl;
        const uint32_t n_mine = c0 < m ? (m - c0 < cpl ? m - c0 : cpl) : 0u;
        for (uint32_t base = 0; base < k; base += 4) {
            const uint32_t j = base + (uint32_t)g;
            const bool on = j < k;
            const uint32_t id = hop_ids[on ? j : 0];
            const uint8_t *codes = reinterpret_cast<const uint8_t *>(a.rows) + (uint64_t)id * a.row_stride;
            float t[CPL];
#pragma unroll
            for (int r = 0; r < CPL; ++r) t[r] = 0.0f;
            if (a.pq_kind == 0) entries<0>(a, qdec, codes, c0, n_mine, t);
            else if (a.pq_kind == 1) entries<1>(a, qdec, codes, c0, n_mine, t);
            else entries<2>(a, qdec, codes, c0, n_mine, t);
            // s = -0.0; s += entry[c] in chunk order: the sum walks through the sixteen lanes of the row
            float s = -0.0f;
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                if (p > 0) s = __shfl_up(s, 1, 64);
                if (sub == p) {
#pragma unroll
                    for (int r = 0; r < CPL; ++r)
                        if ((uint32_t)r < n_mine) s += t[r];
                }
            }
            if (on && sub == 15) hop_scores[j] = a.pq_invert ? -s : s;
        }
        __syncthreads();
    }
};
static bool pq_direct_build_ok(const ScanArgs &a) {
    return a.pq_centroids && (a.pq_chunk == 16 || a.pq_chunk == 8 || a.pq_chunk == 4) && (uint64_t)a.pq_m * a.pq_chunk == a.pq_dim && a.pq_m <= 128 && a.pq_ncent <= 256;
}
template <int CHUNK, int SPLIT, int R>
struct HopPQDirectBuild : HopPQDirect<CHUNK, SPLIT, R> {};
int32_t launch_hnsw_build_pq(hipStream_t st, const ScanArgs &a, const HnswBuildArgs &h, int phase, uint32_t grid, int *per_cu) {
    if (h.lds_query_bytes != 0 && pq_direct_build_ok(a)) {      // the batch's entries are the original vectors themselves (api_hnsw.hip): no table of any kind
        QMX_REQUIRE(h.batch_queries, QMX_ERR_BAD_ARG, "PQ build needs the batch's original vectors");
        if (a.pq_chunk == 16) return launch_hnsw_build_hop<HopPQDirectBuild<16, 4, 2>, HopPQInternalDirect<16>>(st, a, h, phase, grid, per_cu);
        if (a.pq_chunk == 8) return launch_hnsw_build_hop<HopPQDirectBuild<8, 4, 4>, HopPQInternalDirect<8>>(st, a, h, phase, grid, per_cu);
        return launch_hnsw_build_hop<HopPQDirectBuild<4, 4, 8>, HopPQInternalDirect<4>>(st, a, h, phase, grid, per_cu);
    }
    QMX_REQUIRE(a.pq_pair && h.batch_queries, QMX_ERR_BAD_ARG, "PQ build needs the centroid pair table and the batch LUTs");
    return launch_hnsw_build_hop<HopPQBuild, HopPQInternal>(st, a, h, phase, grid, per_cu);
}

// ... over multi-vector points: the searches of an insertion through the LUTs of the new point's ORIGINAL inner vectors (HopMaxSimQ), stored <-> stored pairs
// through score_internal_max_similarity over the pair table
int32_t launch_hnsw_build_maxsim_pq(hipStream_t st, const ScanArgs &a, const HnswBuildArgs &h, int phase, uint32_t grid, int *per_cu) {
    QMX_REQUIRE(a.pq_pair && h.batch_queries && a.mv_offsets, QMX_ERR_BAD_ARG, "multi-vector PQ build needs the centroid pair table, the batch LUTs and the point offsets");
    return launch_hnsw_build_hop<HopMaxSimQ<HopPQ>, HopMaxSimInternal<HopPQInternal>>(st, a, h, phase, grid, per_cu);
}

// pair[c][i][j]: grid (m, ncent), thread j
__global__ __launch_bounds__(256) void pq_pair_table_kernel(PqGeom g, const float *centroids, float *pair) {
    const uint32_t c = blockIdx.x, i = blockIdx.y;
    const uint32_t lo = c * g.chunk, hi = min(lo + g.chunk, g.dim);
    const float *da = centroids + (uint64_t)i * g.dim;
    for (uint32_t j = threadIdx.x; j < g.ncent; j += 256) {
        const float *db = centroids + (uint64_t)j * g.dim;
        float d = -0.0f;
        for (uint32_t k = lo; k < hi; ++k) d += pq_term(g.kind, da[k], db[k]);
        pair[((uint64_t)c * g.ncent + i) * g.ncent + j] = d;
    }
}

// ------------------------------------------------------------------------------------------
// score_internal (:574-618): out[i] = (+/-) sum_c distance(centroid[a_code[c]] chunk c, centroid[b_code[c]] chunk c)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pq_internal_kernel(PqGeom g, const uint8_t *rows, uint64_t row_stride, uint64_t n_rows,
                                                          const float *centroids, const float *pair, const uint32_t *a_ids, const uint32_t *b_ids,
                                                          uint32_t n, float *out, int *err_flag) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t ia = a_ids[i], ib = b_ids[i];
    if (ia >= n_rows || ib >= n_rows) {
        *err_flag = 1;
        return;
    }
    const uint8_t *ca = rows + (uint64_t)ia * row_stride, *cb = rows + (uint64_t)ib * row_stride;
    float s = -0.0f;
    if (pair) {   // the tabulated chunk terms (pq_pair_table_kernel: the loop below, once per centroid pair)
        for (uint32_t c = 0; c < g.m; ++c) s += pair[((uint64_t)c * g.ncent + ca[c]) * g.ncent + cb[c]];
        out[i] = g.invert ? -s : s;
        return;
    }
    for (uint32_t c = 0; c < g.m; ++c) {
        const uint32_t lo = c * g.chunk, hi = min(lo + g.chunk, g.dim);
        const float *da = centroids + (uint64_t)ca[c] * g.dim, *db = centroids + (uint64_t)cb[c] * g.dim;
        float d = -0.0f;
        for (uint32_t k = lo; k < hi; ++k) d += pq_term(g.kind, da[k], db[k]);
        s += d;
    }
    out[i] = g.invert ? -s : s;
}

// ------------------------------------------------------------------------------------------
// encode_vector (:301-329).  grid (ceil(n/256), m): block = 256 vectors x one chunk; the chunk of every
// centroid sits in LDS (broadcast reads), each thread keeps its sub-vector in LDS column `tid`.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pq_encode_kernel(PqGeom g, const float *in, uint64_t n, const float *centroids,
                                                        uint8_t *codes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t c = blockIdx.y;
    const uint32_t lo = c * g.chunk, hi = min(lo + g.chunk, g.dim);
    const uint32_t len = hi - lo;
    float *cen = reinterpret_cast<float *>(smem);                 // [ncent][len]
    float *sub = cen + (size_t)g.ncent * len;                     // [len][256]
    const uint32_t tid = threadIdx.x;
    const uint64_t vec = (uint64_t)blockIdx.x * 256 + tid;
    for (uint32_t i = tid; i < g.ncent * len; i += 256) cen[i] = centroids[(uint64_t)(i / len) * g.dim + lo + i % len];
    if (vec < n)
        for (uint32_t k = 0; k < len; ++k) sub[k * 256 + tid] = in[vec * g.dim + lo + k];
    __syncthreads();
    if (vec >= n) return;
    float min_distance = 3.40282347e+38f;   // f32::MAX
    uint32_t min_index = 0;
    for (uint32_t j = 0; j < g.ncent; ++j) {
        float s = -0.0f;
        for (uint32_t k = 0; k < len; ++k) {
            const float d = sub[k * 256 + tid] - cen[j * len + k];
            s += d * d;                                            // (a - b).powi(2), sequential sum
        }
        if (s < min_distance) {                                    // first minimum wins (:321)
            min_distance = s;
            min_index = j;
        }
    }
    codes[vec * g.m + c] = (uint8_t)min_index;
}

static PqGeom make_geom(uint32_t distance, uint32_t dim, const qmx_pq_params &pq) {
    PqGeom g;
    g.dim = dim;
    g.chunk = pq.chunk_size;
    g.m = (dim + pq.chunk_size - 1) / pq.chunk_size;              // get_vector_division :164-169
    g.ncent = pq.n_centroids;
    g.kind = (distance == QMX_DISTANCE_DOT || distance == QMX_DISTANCE_COSINE) ? 0 : distance == QMX_DISTANCE_MANHATTAN ? 1 : 2;
    g.invert = pq.invert;
    return g;
}

int32_t launch_pq_lut(hipStream_t st, uint32_t distance, uint32_t dim, const qmx_pq_params &pq, const float *d_centroids,
                      const float *d_queries, uint32_t nq, float *d_lut) {
    if (nq == 0) return QMX_OK;
    const PqGeom g = make_geom(distance, dim, pq);
    ::qmx::clear_stale_error();
    if (pq.lut_mfma && g.kind == 0 && pq_lut_lds_bytes(g) <= 64 * 1024 && !option(OPT_PQ_LUT_NO_LDS)) {
        hipLaunchKernelGGL(pq_lut_mfma_lds_kernel, dim3(g.m, (nq + PQ_LUT_QB - 1) / PQ_LUT_QB), dim3(256), pq_lut_lds_bytes(g), st, g, nq, d_queries, d_centroids, d_lut);
    } else if (pq.lut_mfma && g.kind == 0) {
        hipLaunchKernelGGL(pq_lut_mfma_kernel, dim3(g.m, (nq + 31) / 32), dim3(256), 0, st, g, nq, d_queries, d_centroids, d_lut);
    } else {
        hipLaunchKernelGGL(pq_lut_kernel, dim3(g.m, nq), dim3(256), 0, st, g, d_queries, d_centroids, d_lut);
    }
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}


int32_t launch_pq_pair_table(hipStream_t st, uint32_t distance, uint32_t dim, const qmx_pq_params &pq, const float *d_centroids, float *d_pair) {
    const PqGeom g = make_geom(distance, dim, pq);
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(pq_pair_table_kernel, dim3(g.m, g.ncent), dim3(256), 0, st, g, d_centroids, d_pair);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

int32_t launch_pq_internal(hipStream_t st, uint32_t distance, uint32_t dim, const qmx_pq_params &pq, const float *d_centroids,
                           const float *d_pair, const void *rows, uint64_t row_stride, uint64_t n_rows, const uint32_t *a_ids, const uint32_t *b_ids,
                           uint32_t n, float *out, int *err_flag) {
    if (n == 0) return QMX_OK;
    const PqGeom g = make_geom(distance, dim, pq);
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(pq_internal_kernel, dim3((n + 255) / 256), dim3(256), 0, st, g, (const uint8_t *)rows, row_stride, n_rows,
                       d_centroids, option(OPT_NO_PQ_PAIR) ? nullptr : d_pair, a_ids, b_ids, n, out, err_flag);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

int32_t launch_pq_encode(hipStream_t st, uint32_t dim, const qmx_pq_params &pq, const float *d_centroids, const float *d_in,
                         uint64_t n, uint8_t *d_codes) {
    if (n == 0) return QMX_OK;
    const PqGeom g = make_geom(QMX_DISTANCE_EUCLID, dim, pq);
    const size_t lds = ((size_t)g.ncent * g.chunk + (size_t)g.chunk * 256) * sizeof(float);
    QMX_REQUIRE(lds <= 150 * 1024, QMX_ERR_NOT_SUPPORTED, "PQ encode: chunk %u x %u centroids needs %zu B of LDS", g.chunk, g.ncent, lds);
    static thread_local DeviceOnce attr_once;
    if (attr_once.need()) {
        QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pq_encode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr_once.mark();
    }
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(pq_encode_kernel, dim3((uint32_t)((n + 255) / 256), g.m), dim3(256), lds, st, g, d_in, n, d_centroids, d_codes);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

// ------------------------------------------------------------------------------------------
// k-means on a given sample (kmeans.rs:9-169, called per chunk by find_centroids, encoded_vectors_pq.rs:342-407).
// One launch of pq_encode_kernel is update_indexes for EVERY chunk at once (chunks are independent k-means problems);
// pq_train_update_kernel is update_centroids: thread (chunk c, centroid j, component i) walks the sample in row order
// inside each of the `threads` row ranges, f64 partial per range, partials added in range order — the reference's
// CentroidsCounter per rayon thread — then the mean, cast to f32; an empty cluster keeps its centroid (the reference
// re-seeds it randomly: unpinned).  pq_train_diff_kernel: sum(|old - new|) in f32, index order, per chunk -> done flag.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pq_train_update_kernel(PqGeom g, const float *data, uint64_t n, const uint8_t *codes, uint32_t threads,
                                                              const uint8_t *done, float *centroids, float *absdiff) {
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t per_chunk = g.ncent * g.chunk;
    const uint32_t c = (uint32_t)(gid / per_chunk);
    if (c >= g.m) return;
    const uint32_t rem = (uint32_t)(gid % per_chunk), j = rem / g.chunk, i = rem % g.chunk;
    const uint32_t lo = c * g.chunk, hi = min(lo + g.chunk, g.dim);
    if (lo + i >= hi) { absdiff[gid] = 0.0f; return; }
    float *cp = centroids + (uint64_t)j * g.dim + lo + i;
    if (done[c]) { absdiff[gid] = 0.0f; return; }
    double acc = 0.0;
    uint64_t cnt = 0;
    const uint64_t per = n / threads;
    for (uint32_t t = 0; t < threads; ++t) {
        const uint64_t r0 = per * t, r1 = (t + 1 == threads) ? n : per * (t + 1);
        double part = 0.0;
        for (uint64_t r = r0; r < r1; ++r)
            if (codes[r * g.m + c] == j) {
                part += (double)data[r * g.dim + lo + i];
                ++cnt;
            }
        acc += part;
    }
    const float old = *cp;
    const float nv = cnt ? (float)(acc / (double)cnt) : old;
    absdiff[gid] = __builtin_fabsf(old - nv);
    *cp = nv;
}
__global__ void pq_train_diff_kernel(PqGeom g, const float *absdiff, float accuracy, uint8_t *done, uint32_t *iters) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= g.m || done[c]) return;
    const uint32_t lo = c * g.chunk, hi = min(lo + g.chunk, g.dim), w = hi - lo;
    float diff = -0.0f;
    for (uint32_t j = 0; j < g.ncent; ++j)
        for (uint32_t i = 0; i < w; ++i) diff += absdiff[(uint64_t)c * g.ncent * g.chunk + (uint64_t)j * g.chunk + i];
    iters[c] += 1;
    if (diff < accuracy) done[c] = 1;
}

int32_t launch_pq_train(hipStream_t st, uint32_t dim, uint32_t chunk_size, uint32_t n_centroids, const float *d_data, uint64_t n,
                        uint32_t max_iters, float accuracy, uint32_t threads, float *d_centroids, uint32_t *iters_host) {
    qmx_pq_params pq = {};
    pq.chunk_size = chunk_size;
    pq.n_centroids = n_centroids;
    const PqGeom g = make_geom(QMX_DISTANCE_DOT, dim, pq);
    if (threads == 0) threads = 1;
    // first-k init (kmeans.rs:27): centroid j = sample row j, every chunk
    QMX_HIP(hipMemcpyAsync(d_centroids, d_data, (size_t)n_centroids * dim * sizeof(float), hipMemcpyDeviceToDevice, st));
    uint8_t *d_codes = nullptr, *d_done = nullptr;
    float *d_abs = nullptr;
    uint32_t *d_iters = nullptr;
    const size_t n_thr = (size_t)g.m * n_centroids * chunk_size;
    int32_t rc = QMX_OK;
    do {
        if (hipMalloc((void **)&d_codes, (size_t)n * g.m) != hipSuccess || hipMalloc((void **)&d_done, g.m) != hipSuccess ||
            hipMalloc((void **)&d_abs, n_thr * sizeof(float)) != hipSuccess || hipMalloc((void **)&d_iters, (size_t)g.m * 4) != hipSuccess) {
            (void)hipGetLastError();
            set_error("out of device memory for k-means scratch");
            rc = QMX_ERR_OUT_OF_MEMORY;
            break;
        }
        if (hipMemsetAsync(d_done, 0, g.m, st) != hipSuccess || hipMemsetAsync(d_iters, 0, (size_t)g.m * 4, st) != hipSuccess) { rc = QMX_ERR_OTHER; break; }
        std::vector<uint8_t> done(g.m);
        for (uint32_t it = 0; it < max_iters && rc == QMX_OK; ++it) {
            if ((rc = launch_pq_encode(st, dim, pq, d_centroids, d_data, n, d_codes)) != QMX_OK) break;     // update_indexes
            ::qmx::clear_stale_error();
            hipLaunchKernelGGL(pq_train_update_kernel, dim3((uint32_t)((n_thr + 255) / 256)), dim3(256), 0, st, g, d_data, n, d_codes, threads,
                               d_done, d_centroids, d_abs);
            hipLaunchKernelGGL(pq_train_diff_kernel, dim3((g.m + 63) / 64), dim3(64), 0, st, g, d_abs, accuracy, d_done, d_iters);
            if (hipGetLastError() != hipSuccess) { rc = QMX_ERR_OTHER; break; }
            if ((it & 3) == 3 || it + 1 == max_iters) {           // all chunks converged?  (poll every 4 iterations)
                if (hipMemcpyAsync(done.data(), d_done, g.m, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { rc = QMX_ERR_OTHER; break; }
                bool all = true;
                for (uint8_t d : done) all = all && d;
                if (all) break;
            }
        }
        if (rc == QMX_OK && iters_host && hipMemcpyAsync(iters_host, d_iters, (size_t)g.m * 4, hipMemcpyDeviceToHost, st) != hipSuccess) rc = QMX_ERR_OTHER;
        if (rc == QMX_OK && hipStreamSynchronize(st) != hipSuccess) rc = QMX_ERR_OTHER;
    } while (0);
    if (d_codes) (void)hipFree(d_codes);
    if (d_done) (void)hipFree(d_done);
    if (d_abs) (void)hipFree(d_abs);
    if (d_iters) (void)hipFree(d_iters);
    return rc;
}

}  // namespace qmx

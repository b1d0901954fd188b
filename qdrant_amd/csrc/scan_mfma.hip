// scan_mfma.hip — f32 dot / cosine brute-force scan for LARGE query tiles (8..32 queries per pass of the
// stored block) on the f32 matrix cores, with the bits of the x86 AVX2+FMA reference.
//
// Same reference loops as scan_common.hpp (BatchFilteredSearcher::peek_top_iter,
// lib/segment/src/index/hnsw_index/point_scorer.rs:423-472 over dot_similarity_avx,
// lib/segment/src/spaces/simple_avx.rs:167-213).  With >= 8 queries per pass the scan is a dense
// contraction [rows x dim] . [dim x Q]; the VALU kernel is LDS/VALU-bound there (Q = 16: 12.5 ms per
// 30.72 GB scan) while the HBM floor is ~4.9 ms.
//
// Why this is still bit-exact.  dot_similarity_avx keeps 32 independent fma chains per (row, query):
// chain c = 8 r + j (AVX register r, SIMD lane j) accumulates element 32 i + c of every 32-float step i in
// order, then folds the chains with four_way_hsum + hsum256_ps_avx (simple_avx.rs:10-28).
// v_mfma_f32_4x4x1_16b_f32 runs 16 independent 4x4 outer-product blocks with K = 1, i.e. ONE fused
// multiply-add per output element per instruction: D[blk][m][n] = fma(A[blk][m], B[blk][n], C[blk][m][n]).
// Mapping block -> chain, m -> stored row, n -> query, and issuing the instruction once per 32-float step
// makes every accumulator element exactly one AVX chain.  The fold is done with the reference's tree.
//
// Lane roles (lane = 4 * blk + x,  blk = u + 8 * rh):
//   u  = 0..7  which 16-byte piece of the 128-byte step the lane loads (columns 4u .. 4u+3)
//   rh = 0..1  which half of the 8-row tile (rows 4 rh + x)
//   x  = 0..3  A operand: stored row 4 rh + x;   B operand: query 4 g + x  (g = query group)
// One global_load_dwordx4 per lane per step feeds 4 MFMAs (t = 0..3: column 4u + t) per query group;
// the 8 lanes of a row read one full 128-byte line, every fetched byte is used once, as in the VALU scan.
// The query piece comes from the LDS tile with one ds_read_b128 per query group per step.
// acc[g][t] (float4 = rows 4rh .. 4rh+3) of lane (u, rh, x) is chain c = 4u + t of query 4g + x, so
//   r = u >> 1, j = 4 (u & 1) + t  ->  a+b / c+d: lanes u ^ 2 (xor 8), (a+b)+(c+d): u ^ 4 (xor 16),
//   hi128 + lo128: u ^ 1 (xor 4), then (lr0 + lr1) + (lr2 + lr3) over t in-lane.
#include "scan_common.hpp"

namespace qmx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MF_BLOCK = 512;
constexpr int MF_NW = MF_BLOCK / WAVE;

template <int XORMASK>
__device__ __forceinline__ float swz_xor(float v) {
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (XORMASK << 10) | 0x1F));
}
__device__ __forceinline__ float ror8(float v) {   // lane l <-> l ^ 8 inside each row of 16 lanes
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128 /* row_ror:8 */, 0xF, 0xF, true));
}

template <int QT, int UNROLL, bool HAS_IDS, int MODE>
__global__ __launch_bounds__(MF_BLOCK) void scan_f32_mfma_kernel(const ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NG = QT / 4;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.queries);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        const uint32_t n16 = (uint32_t)QT * a.q_stride / 16;
        for (uint32_t i = tid; i < n16; i += MF_BLOCK) dst[i] = src[i];
    }
    __syncthreads();

    const int x = lane & 3;
    const int u = (lane >> 2) & 7;
    const int rh = lane >> 5;
    const unsigned char *rows = reinterpret_cast<const unsigned char *>(a.rows);
    const unsigned char *qbase = smem + (uint32_t)x * a.q_stride + (uint32_t)u * 16;   // + g * 4 * q_stride + i * 128
    const uint32_t gstride = 4u * a.q_stride;
    const int top = (int)a.top;

    uint64_t list[QT];
    uint64_t thr[NG];     // this lane's view: k-th best key of query 4g + x
#pragma unroll
    for (int q = 0; q < QT; ++q) list[q] = 0;
#pragma unroll
    for (int g = 0; g < NG; ++g) thr[g] = 0;

    const uint32_t gw = blockIdx.x * MF_NW + wave;
    const uint32_t tw = gridDim.x * MF_NW;
    const uint64_t n_tiles = (a.n_cand + 7) / 8;
    const uint32_t nseg = a.nseg;

    for (uint64_t tile = gw; tile < n_tiles; tile += tw) {
        // the 4 rows of this lane's half tile (rows 4rh + m); the lane itself streams row 4rh + x
        uint32_t rid[4];
        bool valid[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const uint64_t c = tile * 8 + (uint32_t)(4 * rh + m);
            valid[m] = c < a.n_cand;
            const uint64_t cc = valid[m] ? c : 0;
            uint32_t id = HAS_IDS ? a.ids[cc] : (uint32_t)cc;
            if (HAS_IDS && id >= a.n_rows) {
                if (valid[m]) *a.err_flag = 1;
                id = 0;
                valid[m] = false;
            }
            rid[m] = id;
        }
        const uint32_t my_row = x == 0 ? rid[0] : x == 1 ? rid[1] : x == 2 ? rid[2] : rid[3];
        const unsigned char *rp = rows + (uint64_t)my_row * a.row_stride + (uint32_t)u * 16;

        f32x4 acc[NG][4];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[g][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll UNROLL
        for (uint32_t s = 0; s < nseg; ++s) {
            const uint4 vv = *reinterpret_cast<const uint4 *>(rp + (uint64_t)s * 128);
            const float v0 = __uint_as_float(vv.x), v1 = __uint_as_float(vv.y), v2 = __uint_as_float(vv.z), v3 = __uint_as_float(vv.w);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const uint4 qq = *reinterpret_cast<const uint4 *>(qbase + (uint32_t)g * gstride + s * 128);
                // _mm256_fmadd_ps(v1, v2, sum) of chain 4u + t for rows 4rh..4rh+3 x queries 4g..4g+3
                acc[g][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(v0, __uint_as_float(qq.x), acc[g][0], 0, 0, 0);
                acc[g][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(v1, __uint_as_float(qq.y), acc[g][1], 0, 0, 0);
                acc[g][2] = __builtin_amdgcn_mfma_f32_4x4x1f32(v2, __uint_as_float(qq.z), acc[g][2], 0, 0, 0);
                acc[g][3] = __builtin_amdgcn_mfma_f32_4x4x1f32(v3, __uint_as_float(qq.w), acc[g][3], 0, 0, 0);
            }
        }

        // ---- fold the 32 chains (four_way_hsum, hsum256_ps_avx), scalar tail, top-k ----
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const uint32_t q = (uint32_t)(4 * g + x);
            const float *qf = reinterpret_cast<const float *>(smem + q * a.q_stride);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                float lr[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float c0 = acc[g][t][m];
                    const float s12 = c0 + ror8(c0);                 // sum1 = a + b | sum2 = c + d
                    const float tot = s12 + swz_xor<16>(s12);        // total = sum1 + sum2
                    lr[t] = tot + swz_xor<4>(tot);                   // hi128 + lo128
                }
                float score = (lr[0] + lr[1]) + (lr[2] + lr[3]);
                if (a.tail_start < a.dim) {                          // scalar tail: mul then add (simple_avx.rs:208-211)
                    const float *vf = reinterpret_cast<const float *>(rows + (uint64_t)rid[m] * a.row_stride);
                    for (uint32_t i = a.tail_start; i < a.dim; ++i) score += qf[i] * vf[i];
                }
                const bool mine = u == 0 && valid[m] && q < a.nq;
                if (MODE == SCAN_SCORES) {
                    if (mine) a.scores[(uint64_t)q * a.scores_stride + (tile * 8 + (uint32_t)(4 * rh + m))] = score;
                } else {
                    const uint64_t key = make_key(score, rid[m]);
                    bool c = mine && key > thr[g];
                    if (__ballot(c)) {
                        c = c && a.del.live(rid[m]);
                        uint64_t mask = __ballot(c);
                        while (mask) {
                            const int src = __builtin_ctzll(mask);
                            mask &= mask - 1;
                            const uint64_t nk = readlane_u64(key, src);
                            const int n = src & 3;
#pragma unroll
                            for (int nn = 0; nn < 4; ++nn) {
                                if (n == nn) {
                                    if (nk > readlane_u64(list[4 * g + nn], top - 1)) {
                                        wave_list_insert(list[4 * g + nn], nk, lane);
                                        const uint64_t nt = readlane_u64(list[4 * g + nn], top - 1);
                                        if (x == nn) thr[g] = nt;
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }
    }

    if (MODE == SCAN_SCORES) return;

    // ---- block merge: 8 wave lists -> 1 list per query, one global write per block (as scan_kernel) ----
    __syncthreads();
    uint64_t *lds_keys = reinterpret_cast<uint64_t *>(smem);
    const uint32_t utop = a.top;
#pragma unroll
    for (int q = 0; q < QT; ++q)
        if (lane < top) lds_keys[((uint32_t)wave * QT + q) * utop + lane] = list[q];
    __syncthreads();
    for (uint32_t q = wave; q < a.nq; q += MF_NW) {
        uint64_t merged = 0;
        for (int sw = 0; sw < MF_NW; ++sw) {
            const uint64_t key = lane < top ? lds_keys[((uint32_t)sw * QT + q) * utop + lane] : 0;
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

template <int QT, int UNROLL, bool HAS_IDS, int MODE>
static int32_t launch_mfma_inst(hipStream_t st, const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    size_t lds = (size_t)QT * a.q_stride;
    if (MODE == SCAN_TOPK) {
        const size_t lk = (size_t)MF_NW * QT * a.top * sizeof(uint64_t);
        if (lk > lds) lds = lk;
    }
    lds = (lds + 15) & ~(size_t)15;
    QMX_REQUIRE(lds <= 160 * 1024, QMX_ERR_NOT_SUPPORTED, "query tile needs %zu B of LDS (> 160 KiB)", lds);
    auto kfn = scan_f32_mfma_kernel<QT, UNROLL, HAS_IDS, MODE>;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    int per_cu = 0;
    QMX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, MF_BLOCK, lds));
    if (per_cu < 1) per_cu = 1;
    const uint64_t n_tiles = (a.n_cand + 7) / 8;
    const uint64_t want = (n_tiles + MF_NW - 1) / MF_NW;
    const uint64_t cap = (uint64_t)num_cus * per_cu;
    uint32_t grid = (uint32_t)(want < cap ? want : cap);
    if (grid < 1) grid = 1;
    if (grid_out) {
        if (*grid_out && MODE == SCAN_TOPK && grid > *grid_out) grid = *grid_out;
        *grid_out = grid;
    }
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(MF_BLOCK), lds, st, a);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

template <int QT, int UNROLL>
static int32_t launch_mfma_qt(hipStream_t st, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid) {
    const bool ids = a.ids != nullptr;
    if (mode == SCAN_TOPK)
        return ids ? launch_mfma_inst<QT, UNROLL, true, SCAN_TOPK>(st, a, num_cus, grid)
                   : launch_mfma_inst<QT, UNROLL, false, SCAN_TOPK>(st, a, num_cus, grid);
    return ids ? launch_mfma_inst<QT, UNROLL, true, SCAN_SCORES>(st, a, num_cus, grid)
               : launch_mfma_inst<QT, UNROLL, false, SCAN_SCORES>(st, a, num_cus, grid);
}

// qt in {8, 16, 32}; f32 rows, dot (or cosine on normalised rows), dim >= 32
int32_t launch_scan_f32_mfma(hipStream_t st, int qt, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    switch (qt) {
        case 8: return launch_mfma_qt<8, 4>(st, mode, a, num_cus, grid_out);
        case 16: return launch_mfma_qt<16, 4>(st, mode, a, num_cus, grid_out);
        case 32: return launch_mfma_qt<32, 2>(st, mode, a, num_cus, grid_out);
    }
    set_error("unsupported MFMA query tile %d", qt);
    return QMX_ERR_BAD_ARG;
}

}  // namespace qmx

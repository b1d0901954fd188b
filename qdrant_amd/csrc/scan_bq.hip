// scan_bq.hip — EncodedVectorsBin (binary quantization) on device: every Encoding, QueryEncoding::SameAsStorage | Scalar4bits | Scalar8bits,
// BitsStoreType = u128 (what single-vector segments use, vector_storage/quantized/quantized_vectors/binary/create.rs:34-37).
//
// Reference (lib/quantization/src/encoded_vectors_binary.rs):
//   encode_one_bit_vector   :558-568   bit i = vector[i] > 0.0, little-endian inside u128 words
//   storage size            :829-840 + BitsStoreType::get_storage_size for u128 :412-419  (ceil(dim / 128) * 16 bytes)
//   xor_popcnt (u128)       :288-333 -> cpp/sse.c:54-75 impl_xor_popcnt_sse_uint128
//   calculate_metric        :766-810   xor = popcount(v ^ q); zeros = dim - xor; (Dot | Cosine, !invert) and (L1 | L2, invert)
//                                      -> zeros - xor  (invert derives from the distance, quantized_vectors.rs:232)
//   score_internal          :892-917   the same metric between two stored rows; encode_internal_vector :923-934 = the row itself
// Integer work: the popcount is exact in any order, dim and xor are < 2^24, so the f32 arithmetic of calculate_metric is
// exact: scores are bit-identical to the reference's.
//
// The rows are 16-byte multiples: the lane policies of the dense scan apply unchanged (8 lanes per row, 16 bytes per
// lane per 128-byte step) and with them the tiled scan, pair scoring, rescoring plumbing and the HNSW walk.
#include "hnsw_build.hpp"

namespace qmx {

struct RowBQ {
    static constexpr bool TEMPORAL_ROWS = true;   // 96 / 192-byte rows: most rows end inside a 128-byte line
    static constexpr int NACC = 1;
    static constexpr int NRAUX = 0;
    static constexpr int R16 = 2;
    typedef uint32_t acc_t;
    static __device__ __forceinline__ void row_aux(acc_t (&)[1], const uint4 &) {}
    static __device__ __forceinline__ void mac(acc_t (&a)[NACC], const uint4 &q, const uint4 &v) {
        a[0] += (uint32_t)(__popc(q.x ^ v.x) + __popc(q.y ^ v.y) + __popc(q.z ^ v.z) + __popc(q.w ^ v.w));
    }
    static __device__ __forceinline__ float finish(acc_t (&a)[NACC], acc_t (&)[1], const unsigned char *, const unsigned char *, uint32_t,
                                                   const ScanArgs &args) {
        const float xor_product = (float)reduce8_u32(a[0]);
        const float dim = (float)args.bq_dim;
        const float zeros_count = dim - xor_product;
        return args.bq_flip ? xor_product - zeros_count : zeros_count - xor_product;
    }
};

// QueryEncoding::Scalar4bits / Scalar8bits (encoded_vectors_binary.rs:49-54, 721-756): the query keeps B bits per value, stored as B
// bit planes per u128 word of the row (`encoded_query[B * chunk + b]`); xor_popcnt_scalar (:337-409, cpp/avx2.c / sse.c
// impl_xor_popcnt_scalar{4,8}_*_uint128) = sum over words and planes of popcount(row_word ^ plane_b) << b, and calculate_metric
// (:783-788) divides it by (2^B - 1) in f32 before the same zeros / xor arithmetic.  Integer sums are exact in any order; the one
// division and the two subtractions are done in the reference's order.
template <int B>
struct RowBQScalar {
    static constexpr bool TEMPORAL_ROWS = true;
    static constexpr int NACC = 1;
    static constexpr int NRAUX = 0;
    static constexpr int R16 = 2;
    static constexpr int QPIECES = B;
    typedef uint32_t acc_t;
    static __device__ __forceinline__ void row_aux(acc_t (&)[1], const uint4 &) {}
    static __device__ __forceinline__ void mac(acc_t (&)[NACC], const uint4 &, const uint4 &) {}
    static __device__ __forceinline__ void mac_pieces(acc_t (&a)[NACC], const uint4 (&q)[B], const uint4 &v) {
#pragma unroll
        for (int k = 0; k < B; ++k)
            a[0] += (uint32_t)(__popc(q[k].x ^ v.x) + __popc(q[k].y ^ v.y) + __popc(q[k].z ^ v.z) + __popc(q[k].w ^ v.w)) << k;
    }
    static __device__ __forceinline__ float finish(acc_t (&a)[NACC], acc_t (&)[1], const unsigned char *, const unsigned char *, uint32_t,
                                                   const ScanArgs &args) {
        const float xor_product = (float)reduce8_u32(a[0]) / (float)((1u << B) - 1u);
        const float dim = (float)args.bq_dim;
        const float zeros_count = dim - xor_product;
        return args.bq_flip ? xor_product - zeros_count : zeros_count - xor_product;
    }
};

template <class L>
static int32_t dispatch_bq(const L &l, const ScanArgs &a) {
    if (a.bq_qbits == 4) return l.template row<RowBQScalar<4>>(a);
    if (a.bq_qbits == 8) return l.template row<RowBQScalar<8>>(a);
    return l.template row<RowBQ>(a);
}

// ------------------------------------------------------------------------------------------
// The block scan of BQ rows: ONE LANE PER ROW.  A 1-bit row is 96..192 bytes: with the 8-lanes-per-row layout of the dense
// scan the per-(row, query) work is a cross-lane reduction and a key compare for four xor + popcount per lane (160 M pairs per
// 16-query pass over 10 M rows: 1.8 ms, selection-bound).  Here a lane keeps its own row in registers piece by piece, the
// query pieces come through the scalar unit (wave-uniform addresses: SGPR operands of the xor), and v_bcnt_u32_b32 accumulates: 2 VALU
// instructions per row word and query, no cross-lane traffic; the wave's 64 rows are 64 x row_bytes contiguous bytes.
// Same scores (calculate_metric in the reference's f32 order), same key / list / merge logic as scan_common.hpp.
// ------------------------------------------------------------------------------------------
constexpr int BQR_BLOCK = 256;
constexpr int BQR_NW = BQR_BLOCK / WAVE;

// B = bit planes per query value (1: SameAsStorage; 4 / 8: Scalar4bits / Scalar8bits, planes k of row word p at query piece p * B + k):
// the plane's popcount over the four dwords of a piece is chained through v_bcnt's accumulate operand and enters the row's sum shifted
// by k (xor_popcnt_scalar, encoded_vectors_binary.rs:403-409) - one accumulator per query whatever B.
template <int QT, bool HAS_IDS, int MODE, bool SCALARQ, int B>
__global__ __launch_bounds__(BQR_BLOCK) void bq_rows_kernel(const ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t pieces = a.dim / 16;                       // a.dim = bytes of a stored row (a multiple of 16)
    uint4 *sq = reinterpret_cast<uint4 *>(smem);              // [QT][pieces]
    for (uint32_t i = tid; i < (uint32_t)QT * pieces; i += BQR_BLOCK) {
        const uint32_t q = i / pieces, p = i - q * pieces;
        sq[i] = q < a.nq ? *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(a.queries) + (uint64_t)q * a.q_stride + p * 16)
                         : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();

    const unsigned char *rows = reinterpret_cast<const unsigned char *>(a.rows);
    const int top = (int)a.top;
    const float dimf = (float)a.bq_dim;
    uint64_t list[QT];
    uint64_t floor_key[QT];   // a.gthr: a key known to be <= the query's final k-th best key (the k-th best of a pre-scanned prefix, api_*.hip): keys below it never enter a list
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        list[q] = 0;
        floor_key[q] = (a.gthr && q < (int)a.nq) ? a.gthr[q] : 0;
    }

    const uint64_t stride = (uint64_t)gridDim.x * BQR_BLOCK;
    for (uint64_t base = ((uint64_t)blockIdx.x * BQR_NW + wave) * WAVE; base < a.n_cand; base += stride) {
        const uint64_t c = base + lane;
        bool valid = c < a.n_cand;
        uint32_t id = HAS_IDS ? a.ids[valid ? c : 0] : (uint32_t)(valid ? c : 0);
        if (HAS_IDS && id >= a.n_rows) {
            if (valid) *a.err_flag = 1;
            id = 0;
            valid = false;
        }
        const uint4 *rp = reinterpret_cast<const uint4 *>(rows + (uint64_t)id * a.row_stride);
        uint32_t acc[QT];
#pragma unroll
        for (int q = 0; q < QT; ++q) acc[q] = 0;
#pragma unroll 2
        for (uint32_t p = 0; p < pieces; ++p) {
            const uint4 v = rp[p];
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                if constexpr (B == 1) {
                    uint4 qv;
                    if (SCALARQ) {   // wave-uniform address: the scalar unit fetches the query piece, the xor takes it as an SGPR operand
                        const uint32_t *qg = reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned char *>(a.queries) + (uint64_t)q * a.q_stride) + 4 * p;
                        qv = make_uint4(qg[0], qg[1], qg[2], qg[3]);
                    } else {
                        qv = sq[(uint32_t)q * pieces + p];
                    }
                    acc[q] += (uint32_t)(__popc(v.x ^ qv.x) + __popc(v.y ^ qv.y) + __popc(v.z ^ qv.z) + __popc(v.w ^ qv.w));
                } else {
                    const uint32_t *qg = reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned char *>(a.queries) + (uint64_t)q * a.q_stride) + 4 * B * p;
#pragma unroll
                    for (int k = 0; k < B; ++k) {
                        const uint32_t t = (uint32_t)(__popc(v.x ^ qg[4 * k]) + __popc(v.y ^ qg[4 * k + 1]) + __popc(v.z ^ qg[4 * k + 2]) + __popc(v.w ^ qg[4 * k + 3]));
                        acc[q] += t << k;
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            if (q < (int)a.nq) {
                // calculate_metric (encoded_vectors_binary.rs:766-810)
                const float xor_product = B == 1 ? (float)acc[q] : (float)acc[q] / (float)((1u << B) - 1u);   // :783-788
                const float zeros_count = dimf - xor_product;
                const float score = a.bq_flip ? xor_product - zeros_count : zeros_count - xor_product;
                if (MODE == SCAN_SCORES) {
                    if (valid) a.scores[(uint64_t)q * a.scores_stride + c] = score;
                } else {
                    const uint64_t key = make_key(score, id);
                    const uint64_t thr = readlane_u64(list[q], top - 1);
                    bool cnd = valid && key > thr && key >= floor_key[q];
                    if (__ballot(cnd)) {
                        cnd = cnd && a.del.live(id) && (!a.key_bound || key < a.key_bound[q]);
                        uint64_t m = __ballot(cnd);
                        while (m) {
                            const int src = __builtin_ctzll(m);
                            m &= m - 1;
                            const uint64_t nk = readlane_u64(key, src);
                            if (nk > readlane_u64(list[q], top - 1)) wave_list_insert(list[q], nk, lane);
                        }
                    }
                }
            }
        }
    }
    if (MODE == SCAN_SCORES) return;

    // ---- block merge: 4 wave lists -> 1 list per query, one global write per block ----
    __syncthreads();
    uint64_t *lds_keys = reinterpret_cast<uint64_t *>(smem);
    const uint32_t utop = a.top;
#pragma unroll
    for (int q = 0; q < QT; ++q)
        if (lane < top) lds_keys[((uint32_t)wave * QT + q) * utop + lane] = list[q];
    __syncthreads();
    for (uint32_t q = wave; q < a.nq; q += BQR_NW) {
        uint64_t merged = 0;
        for (int sw = 0; sw < BQR_NW; ++sw) {
            const uint64_t key = lane < top ? lds_keys[((uint32_t)sw * QT + q) * utop + lane] : 0;
            uint64_t mk = __ballot(key > readlane_u64(merged, top - 1));
            while (mk) {
                const int src = __builtin_ctzll(mk);
                mk &= mk - 1;
                const uint64_t nk = readlane_u64(key, src);
                if (nk > readlane_u64(merged, top - 1)) wave_list_insert(merged, nk, lane);
            }
        }
        if (lane < top) a.partial[((uint64_t)blockIdx.x * a.partial_qt + q) * utop + lane] = merged;
    }
}

template <int QT, bool HAS_IDS, int MODE, int B = 1>
static int32_t launch_bq_rows_inst(hipStream_t st, const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    size_t lds = (size_t)QT * a.dim;
    if (MODE == SCAN_TOPK) lds = std::max(lds, (size_t)BQR_NW * QT * a.top * sizeof(uint64_t));
    lds = (lds + 15) & ~(size_t)15;
    QMX_REQUIRE(lds <= 64 * 1024, QMX_ERR_NOT_SUPPORTED, "BQ query tile needs %zu B of LDS", lds);
    const uint64_t want = (a.n_cand + BQR_BLOCK - 1) / BQR_BLOCK;
    const uint64_t cap = (uint64_t)num_cus * 8;
    uint32_t grid = (uint32_t)(want < cap ? want : cap);
    if (grid < 1) grid = 1;
    if (grid_out) {
        if (*grid_out && MODE == SCAN_TOPK && grid > *grid_out) grid = *grid_out;
        *grid_out = grid;
    }
    ::qmx::clear_stale_error();
    QMX_NOTE_KERNEL((bq_rows_kernel<QT, HAS_IDS, MODE, true, B>));
    hipLaunchKernelGGL((bq_rows_kernel<QT, HAS_IDS, MODE, true, B>), dim3(grid), dim3(BQR_BLOCK), lds, st, a);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}
template <int QT, int B = 1>
static int32_t launch_bq_rows_qt(hipStream_t st, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid) {
    const bool ids = a.ids != nullptr;
    if (mode == SCAN_TOPK) return ids ? launch_bq_rows_inst<QT, true, SCAN_TOPK, B>(st, a, num_cus, grid) : launch_bq_rows_inst<QT, false, SCAN_TOPK, B>(st, a, num_cus, grid);
    return ids ? launch_bq_rows_inst<QT, true, SCAN_SCORES, B>(st, a, num_cus, grid) : launch_bq_rows_inst<QT, false, SCAN_SCORES, B>(st, a, num_cus, grid);
}
template <int B>
static int32_t launch_bq_rows_planes(hipStream_t st, int qt, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid) {
    switch (qt) {
        case 1: return launch_bq_rows_qt<1, B>(st, mode, a, num_cus, grid);
        case 2: return launch_bq_rows_qt<2, B>(st, mode, a, num_cus, grid);
        case 4: return launch_bq_rows_qt<4, B>(st, mode, a, num_cus, grid);
        case 8: return launch_bq_rows_qt<8, B>(st, mode, a, num_cus, grid);
        case 16: return launch_bq_rows_qt<16, B>(st, mode, a, num_cus, grid);
    }
    set_error("unsupported BQ query tile %d", qt);
    return QMX_ERR_BAD_ARG;
}

int32_t launch_scan_bq(hipStream_t st, int qt, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    // measured on 10 M x 768 / 1536 bits: 1..2 queries 0.19 / 0.34 ms on the 8-lanes-per-row layout (0.24 / 0.60 here), 4 queries
    // 0.27 / 0.64 ms here (0.44 / 0.63 there), 16 queries 0.90 / 1.21 ms here (1.8 / 2.2 there)
    // scalar-encoded queries: one lane per row (the 8-lanes-per-row layout re-reads B query pieces from LDS per row piece: 10 M x 768
    // bits, 8 planes: 0.38 / 1.44 / 5.25 ms for 1 / 4 / 16 queries there, 0.28 / 0.73 / 2.64 ms here = 2/3 of the VALU rate of
    // xor + bcnt per plane dword); a single query over rows of more than one line stays on the 8-lane layout (1536 bits: 0.63 vs 0.76 ms)
    const bool rows_kernel = !(qt == 1 && a.dim > 128);
    if (a.bq_qbits == 4 && rows_kernel) return launch_bq_rows_planes<4>(st, qt, mode, a, num_cus, grid_out);
    if (a.bq_qbits == 8 && rows_kernel) return launch_bq_rows_planes<8>(st, qt, mode, a, num_cus, grid_out);
    if (qt <= 2 || a.bq_qbits > 1) return dispatch_bq(ScanLauncher{st, qt, mode, num_cus, grid_out}, a);
    switch (qt) {
        case 4: return launch_bq_rows_qt<4>(st, mode, a, num_cus, grid_out);
        case 8: return launch_bq_rows_qt<8>(st, mode, a, num_cus, grid_out);
        case 16: return launch_bq_rows_qt<16>(st, mode, a, num_cus, grid_out);
    }
    set_error("unsupported BQ query tile %d", qt);
    return QMX_ERR_BAD_ARG;
}
int32_t launch_pairs_bq(hipStream_t st, const ScanArgs &a, const PairSel &sel, uint64_t n_items, int num_cus) {
    return dispatch_bq(PairLauncher{st, sel, n_items, num_cus}, a);
}
int32_t launch_hnsw_bq(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu) {
    return dispatch_bq(HnswLauncher{st, &h, grid, per_cu}, a);
}
int32_t launch_hnsw_custom_bq(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu) {
    return dispatch_bq(HnswCustomLauncher{st, &h, grid, per_cu}, a);
}
int32_t launch_hnsw_maxsim_bq(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu) {
    return dispatch_bq(HnswMaxSimLauncher{st, &h, grid, per_cu}, a);
}
// device HNSW build through the BQ scorer: a stored bit row IS its internal query (encode_internal_vector :923-934), one bit per value
// whatever the segment's QueryEncoding (score_internal :892-917)
int32_t launch_hnsw_build_bq(hipStream_t st, const ScanArgs &a, const HnswBuildArgs &h, int phase, uint32_t grid, int *per_cu) {
    return HnswBuildLauncher{st, &h, phase, grid, per_cu}.template row<RowBQ>(a);
}
int32_t launch_hnsw_build_maxsim_bq(hipStream_t st, const ScanArgs &a, const HnswBuildArgs &h, int phase, uint32_t grid, int *per_cu) {
    return HnswBuildMaxSimLauncher{st, &h, phase, grid, per_cu}.template row<RowBQ>(a);
}

// encode_vector for a batch (encoded_vectors_binary.rs:535-672): in [n][dim] f32 -> out [n][out_stride] bytes; one thread per
// output dword.  encoding 0: bit i = v[i] > 0.  1 (TwoBits): bit i = b1(v[i]), bit dim + i = b2(v[i]).  2 (OneAndHalfBits): bit
// i = b1(v[i]), bit dim + k = b2(v[2k]) | b2(v[2k+1]).  (b1, b2) = encode_two_bits_value (:626-672).
__device__ __forceinline__ void bq_two_bits(float value, const float *mean, const float *stddev, uint32_t i, bool *b1, bool *b2) {
    if (!mean || !stddev) { *b1 = *b2 = value > 0.0f; return; }                  // no stats: (true, true) / (false, false)
    const float sd = stddev[i];
    if (sd < 1.1920929e-07f) { *b1 = value > 0.0f; *b2 = false; return; }        // f32::EPSILON: regular BQ with the zero comparison
    const float v_z = (value - mean[i]) / sd;
    const float SIGMAS = 2.0f / 3.0f;
    if (v_z <= -SIGMAS) { *b1 = false; *b2 = false; }
    else if (v_z < SIGMAS) { *b1 = true; *b2 = false; }
    else { *b1 = true; *b2 = true; }
}
__global__ __launch_bounds__(256) void bq_encode_kernel(const float *in, uint64_t n, uint32_t dim, uint32_t encoding, const float *mean,
                                                        const float *stddev, uint32_t row_bytes, uint8_t *out, uint64_t out_stride) {
    const uint32_t words = row_bytes / 4;
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t r = gid / words;
    const uint32_t w = (uint32_t)(gid % words);
    if (r >= n) return;
    const float *v = in + r * dim;
    uint32_t bits = 0;
    for (uint32_t b = 0; b < 32; ++b) {
        const uint32_t j = w * 32 + b;
        bool on = false;
        if (encoding == 0) {
            on = j < dim && v[j] > 0.0f;
        } else if (j < dim) {
            bool b1, b2;
            bq_two_bits(v[j], mean, stddev, j, &b1, &b2);
            on = b1;
        } else if (encoding == 1) {
            const uint32_t i = j - dim;
            if (i < dim) { bool b1, b2; bq_two_bits(v[i], mean, stddev, i, &b1, &b2); on = b2; }
        } else {
            const uint32_t k = j - dim;
            for (uint32_t i = 2 * k; i < 2 * k + 2 && i < dim; ++i) { bool b1, b2; bq_two_bits(v[i], mean, stddev, i, &b1, &b2); on = on || b2; }
        }
        if (on) bits |= 1u << b;
    }
    *reinterpret_cast<uint32_t *>(out + r * out_stride + (uint64_t)w * 4) = bits;
}
uint64_t bq_row_bytes(uint32_t dim, uint32_t encoding) {   // get_quantized_vector_size_from_params::<u128> (:829-840)
    uint64_t ext = encoding == 0 ? dim : encoding == 1 ? 2ull * dim : (3ull * dim + 1) / 2;
    if (ext < 1) ext = 1;
    return (ext + 127) / 128 * 16;
}
// encode_scalar_query_vector + _encode_scalar_query_vector (encoded_vectors_binary.rs:692-756) for a batch of queries: one block per
// query.  The query is extended as the row encoding extends the row (TwoBits: the values twice; OneAndHalfBits: the values, then the
// max of each pair), quantised to `bits` bits over [-max_abs, max_abs] in the reference's f32 steps (v - min, / delta, round half
// away, % 2^bits), and stored as bit planes: dword `part` of u128 word `bits * chunk + b` holds bit b of values chunk * 128 + part * 32 + e.
// Behind the planes (qbytes_off; 0 = not wanted) the same values as bytes for the matrix-core scan (scan_sq_mfma.hip BqOps): per 16-byte row piece 8
// pieces, piece j = the values of the row bits = j mod 8 in row-byte order, 8-bit values less 128 (i8); their sum goes to the entry's aux block.
__global__ __launch_bounds__(256) void bq_encode_scalar_query_kernel(const float *in, uint32_t dim, uint32_t encoding, uint32_t bits, uint8_t *out,
                                                                     uint32_t out_stride, uint32_t qbytes_off, uint32_t aux_off, uint32_t body) {
    __shared__ float red[256];
    __shared__ uint32_t value_sum;
    if (threadIdx.x == 0) value_sum = 0;
    const float *q = in + (uint64_t)blockIdx.x * dim;
    const uint32_t ext = encoding == QMX_BQ_TWO_BITS ? 2 * dim : encoding == QMX_BQ_ONE_AND_HALF_BITS ? dim + (dim + 1) / 2 : dim;
    auto value = [&](uint32_t i) -> float {
        if (i < dim) return q[i];
        if (encoding == QMX_BQ_TWO_BITS) return q[i - dim];
        const uint32_t k = 2 * (i - dim);
        return k + 1 < dim ? fmaxf(q[k], q[k + 1]) : q[k];        // f32::max: the non-NaN operand
    };
    float m = 0.0f;
    for (uint32_t i = threadIdx.x; i < ext; i += 256) m = fmaxf(m, fabsf(value(i)));   // fold(0.0, f32::max)
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    const float max_abs = red[0];
    const float mn = -max_abs, mx = max_abs;
    const uint32_t ranges = (1u << bits) - 1u;
    const float delta = (mx - mn) / (float)ranges;
    const uint32_t n_chunks = (ext > 0 ? ext : 1) / 128 + (((ext > 0 ? ext : 1) % 128) ? 1 : 0);   // get_storage_size(len.max(1))
    const uint32_t n_dwords = n_chunks * bits * 4;
    uint32_t *dst = reinterpret_cast<uint32_t *>(out + (uint64_t)blockIdx.x * out_stride);
    for (uint32_t w = threadIdx.x; w < n_dwords; w += 256) {
        const uint32_t chunk = w / (bits * 4), b = (w / 4) % bits, part = w % 4;
        uint32_t word = 0;
        for (uint32_t e = 0; e < 32; ++e) {
            const uint32_t i = chunk * 128 + part * 32 + e;
            if (i >= ext) break;
            const float shifted = value(i) - mn;
            const float delted = delta > 1.1920929e-07f ? shifted / delta : 0.0f;     // f32::EPSILON
            const float r = roundf(delted);
            const uint32_t rounded = !(r >= 0.0f) ? 0u : (r >= 1073741824.0f ? 1073741824u : (uint32_t)r);   // `as usize`: NaN / negative -> 0
            const uint32_t quantized = rounded % (ranges + 1u);
            word |= ((quantized >> b) & 1u) << e;
        }
        dst[w] = word;
    }
    if (!qbytes_off) return;
    uint8_t *entry = out + (uint64_t)blockIdx.x * out_stride;
    uint32_t mine = 0;
    for (uint32_t i = threadIdx.x; i < body * 8; i += 256) {
        uint32_t quantized = 0;
        if (i < ext) {
            const float shifted = value(i) - mn;
            const float delted = delta > 1.1920929e-07f ? shifted / delta : 0.0f;
            const float r = roundf(delted);
            const uint32_t rounded = !(r >= 0.0f) ? 0u : (r >= 1073741824.0f ? 1073741824u : (uint32_t)r);
            quantized = rounded % (ranges + 1u);
            mine += quantized;
        }
        const uint32_t piece = i / 128, d = i % 128;
        entry[qbytes_off + ((size_t)byte_form_slot(piece) * 8 + d % 8) * 16 + d / 8] = i < ext ? (uint8_t)(int8_t)((int32_t)quantized - (bits == 8 ? 128 : 0)) : 0;
    }
    atomicAdd(&value_sum, mine);
    __syncthreads();
    if (threadIdx.x == 0) reinterpret_cast<QueryAux *>(entry + aux_off)->pad[1] = value_sum;
}
int32_t launch_bq_encode_scalar_query(hipStream_t st, const float *d_in, uint32_t nq, uint32_t dim, uint32_t encoding, uint32_t bits, uint8_t *d_out,
                                      uint32_t out_stride, uint32_t qbytes_off, uint32_t aux_off, uint32_t body) {
    if (nq == 0) return QMX_OK;
    QMX_REQUIRE(bits == 4 || bits == 8, QMX_ERR_BAD_ARG, "scalar query encoding of %u bits", bits);
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(bq_encode_scalar_query_kernel, dim3(nq), dim3(256), 0, st, d_in, dim, encoding, bits, d_out, out_stride, qbytes_off, aux_off, body);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

// VectorStats::build (vector_stats.rs:27-117): one thread per dimension runs the streaming Welford update over the rows in order (f64 mean / m2, the
// division by the running count included - the reference's bits), rows read 8 ahead; adjacent threads read adjacent floats of a row.
__global__ __launch_bounds__(64) void vector_stats_kernel(const float *rows, uint64_t row_stride_f, uint64_t n, uint32_t dim, float *mn, float *mx, float *mean,
                                                          float *stddev) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= dim) return;
    float lo = 3.40282347e+38f, hi = -3.40282347e+38f;
    double m = 0.0, m2 = 0.0;
    const float *p = rows + d;
    uint64_t r = 0;
    for (; r + 8 <= n; r += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(p + (r + k) * row_stride_f);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double value = (double)v[k];
            lo = v[k] < lo ? v[k] : lo;
            hi = v[k] > hi ? v[k] : hi;
            const double delta = value - m;
            m += delta / (double)(r + k + 1);
            m2 += delta * (value - m);
        }
    }
    for (; r < n; ++r) {
        const float x = p[r * row_stride_f];
        const double value = (double)x;
        lo = x < lo ? x : lo;
        hi = x > hi ? x : hi;
        const double delta = value - m;
        m += delta / (double)(r + 1);
        m2 += delta * (value - m);
    }
    mn[d] = lo;
    mx[d] = hi;
    mean[d] = (float)m;
    stddev[d] = n > 1 ? (float)sqrt(m2 / (double)(n - 1)) : 0.0f;
}
int32_t launch_vector_stats(hipStream_t st, const float *d_rows, uint64_t row_stride_bytes, uint64_t n, uint32_t dim, float *d_min, float *d_max, float *d_mean,
                            float *d_stddev) {
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(vector_stats_kernel, dim3((dim + 63) / 64), dim3(64), 0, st, d_rows, row_stride_bytes / 4, n, dim, d_min, d_max, d_mean, d_stddev);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

int32_t launch_bq_encode(hipStream_t st, const float *d_in, uint64_t n, uint32_t dim, uint32_t encoding, const float *d_mean, const float *d_stddev,
                         uint8_t *d_out, uint64_t out_stride) {
    if (n == 0) return QMX_OK;
    const uint32_t row_bytes = (uint32_t)bq_row_bytes(dim, encoding);
    const uint64_t total = n * (row_bytes / 4);
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(bq_encode_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, d_in, n, dim, encoding, d_mean, d_stddev, row_bytes,
                       d_out, out_stride);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

}  // namespace qmx

// scan_sq_mfma.hip — EncodedVectorsU8 (scalar int8) brute-force scan for LARGE query tiles (8..32 queries per
// pass) on the int8 matrix cores.
//
// Same reference loops as scan_quant.hip: BatchFilteredSearcher::peek_top_iter
// (lib/segment/src/index/hnsw_index/point_scorer.rs:423-472) over EncodedVectorsU8::score_point_avx
// (lib/quantization/src/encoded_vectors_u8.rs:471-490 -> cpp/avx2.c:25-63 impl_score_dot_avx) and
// postprocess_score (:100-103).  Dot / cosine / euclid (all three are the integer dot of the codes, only
// multiplier and offsets differ, :205-221); Manhattan (sad) stays on the VALU kernel.
//
// Exactness.  Codes are <= 127, so every product and every i32 partial sum is exact in any order.  The AVX2
// leaf converts its 8 i32 lane sums to f32 and adds them (HSUM256_PS); while 127^2 * actual_dim < 2^24 those
// f32 adds are exact too, so the leaf's result is (float)(total integer dot) whatever the association — the
// same argument RowSQ<false, false> rests on.  Above that bound (actual_dim >= 1041) the f32 adds may round and
// the VALU kernel, which keeps the 8 lanes apart, is used instead (`sq_mfma_ok`).
//
// Mapping: v_mfma_i32_16x16x64_i8, one instruction = 16 stored rows x 16 queries x 64 code bytes.
//   lane = m + 16 kg:  A = bytes [64 s + 16 kg, +16) of row m (one global_load_dwordx4 per lane per step; the 4
//   lanes of a row read 64 contiguous bytes, the next step the other half of the line), B = the same bytes of
//   query n = lane % 16 from the LDS tile (ds_read_b128), D[r] = rows 4 (lane / 16) + r x query lane % 16.
//   A wave therefore finishes 16 rows x QW queries with QW / 16 accumulators of 4 registers.
// HBM-bound: 4 + actual_dim bytes per scored row (772 B at d = 768), 12 MFMAs per 16 rows x 16 queries.
#include "scan_common.hpp"

namespace qmx {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// What differs between the element types that share this kernel (64 row bytes per lane-step, 16 rows x 16 queries per MFMA)
// byte offset, inside a query entry, of the operand this kernel reads (Tq1Ops: the i8 form behind the bit planes); 0 unless the Ops say otherwise
// whether finish() wants the number of set bits of the row (counted from the pieces the wave reads anyway)
template <class Ops, class = void> struct ops_row_ones { static constexpr bool value = false; };
template <class Ops> struct ops_row_ones<Ops, decltype((void)Ops::ROW_ONES)> { static constexpr bool value = Ops::ROW_ONES; };
// where the 16 bytes x pieces of lane group kg sit inside a step of the query entry: kg * QSTEP / 4 unless the Ops permute them
template <class Ops, class = void> struct ops_kg_off { static __device__ __forceinline__ uint32_t get(int kg) { return (uint32_t)kg * (Ops::QSTEP / 4); } };
template <class Ops> struct ops_kg_off<Ops, decltype((void)&Ops::kg_off)> { static __device__ __forceinline__ uint32_t get(int kg) { return Ops::kg_off(kg); } };
template <class Ops, class = void> struct ops_query_off { static __device__ __forceinline__ uint32_t get(const ScanArgs &) { return 0; } };
template <class Ops> struct ops_query_off<Ops, decltype((void)&Ops::query_off)> { static __device__ __forceinline__ uint32_t get(const ScanArgs &a) { return Ops::query_off(a); } };

__device__ __forceinline__ i32x4 mfma_i8(const uint4 &x, const uint4 &y, i32x4 c) {
    return __builtin_amdgcn_mfma_i32_16x16x64_i8((i32x4){(int)x.x, (int)x.y, (int)x.z, (int)x.w}, (i32x4){(int)y.x, (int)y.y, (int)y.z, (int)y.w}, c, 0, 0, 0);
}
// An Ops type says: NA accumulators per 16-query group, QSTEP query bytes per 64-byte row step (the lane's share: QSTEP / 4 at kg * QSTEP / 4),
// how a lane's 16 row bytes are decoded (once per step, not once per query group) and multiplied, and how a score is finished.
struct SqOps {     // EncodedVectorsU8: exact integer dot, then postprocess_score
    typedef i32x4 acc_t;
    static constexpr int NA = 1;
    static constexpr uint32_t QSTEP = 64;
    typedef uint4 dec_t;
    static __device__ __forceinline__ uint32_t body_bytes(const ScanArgs &a) { return a.dim; }   // actual_dim code bytes
    static __device__ __forceinline__ void decode(const uint4 &x, dec_t &d) { d = x; }
    static __device__ __forceinline__ void mac(const dec_t &d, const unsigned char *qp, acc_t (&acc)[NA]) {
        acc[0] = mfma_i8(d, *reinterpret_cast<const uint4 *>(qp), acc[0]);
    }
    static __device__ __forceinline__ float row_aux(const ScanArgs &a, uint32_t rid) { return a.row_offsets[rid]; }
    // multiplier * dot + query_offset + vector_offset, left to right, not fused (encoded_vectors_u8.rs:100-103)
    struct qc_t { float q_off; };     // what finish() needs of the lane's query, read from the LDS tile once per launch
    static __device__ __forceinline__ qc_t load_qc(const ScanArgs &a, const unsigned char *q_entry) {
        return qc_t{reinterpret_cast<const QueryAux *>(q_entry + a.aux_off)->f0};
    }
    static __device__ __forceinline__ float finish(const ScanArgs &a, const acc_t (&acc)[NA], int r, const qc_t &qc, const unsigned char *, uint32_t,
                                                   float v_off, uint32_t) {
        const float m1 = a.sq_multiplier * (float)acc[0][r];
        const float mq = m1 + qc.q_off;
        return mq + v_off;
    }
};
// TurboQuant 4 / 2 bits (scan_tq.hip: the same integer arithmetic on the VALU): a lane's 16 code bytes are 32 / 64 dims; decoded to signed codebook
// bytes they are 2 / 4 operand registers (even / odd dims; dims = j mod 4), each multiplied with the low and the high half of the query
// (q_signed = 128 high + low): 4 / 8 matrix instructions per step and group, two i32 accumulators.  The query entry keeps its scan_tq.hip layout:
// per 16-byte row piece [low pieces][high pieces], i.e. 64 / 128 query bytes per lane and step.
template <int BITS, bool L2>
struct TqOps {
    typedef i32x4 acc_t;
    static constexpr int NA = 2;                      // sum low * c, sum high * c
    static constexpr int NP = BITS == 4 ? 2 : 4;      // operand registers a decoded piece fills
    static constexpr uint32_t QSTEP = 64 * 2 * NP;    // query bytes per 64-byte row step
    struct dec_t { uint4 c[NP]; };
    static __device__ __forceinline__ uint32_t body_bytes(const ScanArgs &a) { return a.dim; }   // code bytes of a device row (16-byte multiple)
    static __device__ __forceinline__ uint32_t lut4(uint32_t sel) {
        const uint32_t s = sel & 0x07070707u;
        const uint32_t lo = __builtin_amdgcn_perm(0xFAEEE1D4u, 0xC5B49F80u, s), hi = __builtin_amdgcn_perm(0x7F614C3Bu, 0x2C1F1206u, s);
        return __builtin_amdgcn_perm(hi, lo, ((sel >> 1) & 0x04040404u) | 0x03020100u);   // byte i: lo's, or hi's where bit 3 of the code is set (as tq4_lookup, tq_policies.hpp)
    }
    static __device__ __forceinline__ void decode(const uint4 &x, dec_t &d) {
        const uint32_t v[4] = {x.x, x.y, x.z, x.w};
        uint32_t o[NP][4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (BITS == 4) {
                o[0][w] = lut4(v[w] & 0x0F0F0F0Fu);
                o[1][w] = lut4((v[w] >> 4) & 0x0F0F0F0Fu);
            } else {
#pragma unroll
                for (int j = 0; j < NP; ++j) o[j][w] = __builtin_amdgcn_perm(0u, 0x7F26DA80u, (v[w] >> (2 * j)) & 0x03030303u);
            }
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) d.c[j] = make_uint4(o[j][0], o[j][1], o[j][2], o[j][3]);
    }
    static __device__ __forceinline__ void mac(const dec_t &d, const unsigned char *qp, acc_t (&acc)[NA]) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            acc[0] = mfma_i8(d.c[j], *reinterpret_cast<const uint4 *>(qp + j * 16), acc[0]);
            acc[1] = mfma_i8(d.c[j], *reinterpret_cast<const uint4 *>(qp + (NP + j) * 16), acc[1]);
        }
    }
    static __device__ __forceinline__ float row_aux(const ScanArgs &a, uint32_t rid) { return a.tq_sf[rid]; }
    struct qc_t { float f0, ec, qlsq; };
    static __device__ __forceinline__ qc_t load_qc(const ScanArgs &a, const unsigned char *q_entry) {
        const QueryAux *aux = reinterpret_cast<const QueryAux *>(q_entry + a.aux_off);
        const float ql = __uint_as_float(aux->pad[0]);
        return qc_t{aux->f0, __uint_as_float(aux->pad[3]), ql * ql};
    }
    static __device__ __forceinline__ float finish(const ScanArgs &a, const acc_t (&acc)[NA], int r, const qc_t &qc, const unsigned char *, uint32_t rid,
                                                   float sf, uint32_t) {
        // low + 128 high: |.| < 2^31 below ~2000 coordinates (api_query.hip sets tq_i32): one v_cvt instead of the i64 -> f32 sequence, the same value
        const float sumf = a.tq_i32 ? (float)(acc[0][r] + 128 * acc[1][r]) : (float)((int64_t)acc[0][r] + 128 * (int64_t)acc[1][r]);
        const float dot = qc.f0 * sumf + qc.ec;
        float score;
        if (L2) {
            const float l2 = a.tq_l2[rid];
            const float y = l2 * l2, z = (2.0f * dot) * sf;
            score = (qc.qlsq + y) - z;
        } else {
            score = dot * sf;
        }
        return a.tq_invert ? -score : score;
    }
};
// TurboQuant 1 bit (and 1.5): a lane's 16 row bytes are 128 dims; bit j of every byte -> operand register j (0 / 1 bytes, the dims = j mod 8 in
// row-byte order) against the i8 form of the query (scan_tq.hip tq_query_encode_kernel): v . q = sum over set bits of q, 8 matrix instructions per
// step and group (16 with the two halves of a 16-bit TQ+ query: NA = 2; +-32767 does not fit two signed digits, so the low one is stored
// less 128 and the row's count of set bits, ROW_ONES, pays it back).  score = scale * (2 v . q - sum q), as RowTQ1.
template <int NA_, bool L2>
struct Tq1Ops {
    typedef i32x4 acc_t;
    static constexpr int NA = NA_;
    static constexpr uint32_t QSTEP = 4 * 128 * NA_;
    static constexpr bool ROW_ONES = NA_ == 2;
    struct dec_t { uint4 c[8]; };
    static __device__ __forceinline__ uint32_t body_bytes(const ScanArgs &a) { return a.dim; }
    static __device__ __forceinline__ uint32_t query_off(const ScanArgs &a) { return a.tq_qbytes_off; }
    // 8-bit values: a lane group's 8 pieces are 128 bytes; groups 0 / 1 (and 2 / 3) are read in the same LDS cycle, so they sit 256 bytes apart
    // (order 0, 2, 1, 3 inside the step - tq_piece_slot below is what the encoders use)
    static __device__ __forceinline__ uint32_t kg_off(int kg) { return NA_ == 1 ? (uint32_t)(((kg & 1) << 1) | (kg >> 1)) * 128u : (uint32_t)kg * 256u; }
    static __device__ __forceinline__ void decode(const uint4 &x, dec_t &d) {
        const uint32_t v[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int j = 0; j < 8; ++j)
            d.c[j] = make_uint4((v[0] >> j) & 0x01010101u, (v[1] >> j) & 0x01010101u, (v[2] >> j) & 0x01010101u, (v[3] >> j) & 0x01010101u);
    }
    static __device__ __forceinline__ void mac(const dec_t &d, const unsigned char *qp, acc_t (&acc)[NA]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[0] = mfma_i8(d.c[j], *reinterpret_cast<const uint4 *>(qp + j * 16), acc[0]);
            if (NA == 2) acc[NA - 1] = mfma_i8(d.c[j], *reinterpret_cast<const uint4 *>(qp + (8 + j) * 16), acc[NA - 1]);
        }
    }
    static __device__ __forceinline__ float row_aux(const ScanArgs &a, uint32_t rid) { return a.tq_sf[rid]; }
    struct qc_t { float f0, ec, qlsq; int64_t sum_q; };
    static __device__ __forceinline__ qc_t load_qc(const ScanArgs &a, const unsigned char *q_entry) {
        const QueryAux *aux = reinterpret_cast<const QueryAux *>(q_entry + a.aux_off);
        const float ql = __uint_as_float(aux->pad[0]);
        return qc_t{aux->f0, __uint_as_float(aux->pad[3]), ql * ql, (int64_t)(((uint64_t)aux->pad[2] << 32) | aux->pad[1])};
    }
    static __device__ __forceinline__ float finish(const ScanArgs &a, const acc_t (&acc)[NA], int r, const qc_t &qc, const unsigned char *, uint32_t rid,
                                                   float sf, uint32_t ones) {
        float signed_dot;
        if (a.tq_i32) {      // |2 v.q - sum q| < 2^31 (api_query.hip): 32-bit arithmetic, one v_cvt; the same value as the i64 form
            int32_t v_dot_q = acc[0][r];
            if (NA == 2) v_dot_q += 256 * acc[NA - 1][r] + 128 * (int32_t)ones;          // q = 256 (q >> 8) + ((q & 255) - 128) + 128
            signed_dot = (float)(2 * v_dot_q - (int32_t)qc.sum_q);
        } else {
            int64_t v_dot_q = (int64_t)acc[0][r];
            if (NA == 2) v_dot_q += 256 * (int64_t)acc[NA - 1][r] + 128 * (int64_t)ones;
            signed_dot = (float)(2 * v_dot_q - qc.sum_q);
        }
        const float dot = qc.f0 * signed_dot + qc.ec;
        float score;
        if (L2) {
            const float l2 = a.tq_l2[rid];
            const float y = l2 * l2, z = (2.0f * dot) * sf;
            score = (qc.qlsq + y) - z;
        } else {
            score = dot * sf;
        }
        return a.tq_invert ? -score : score;
    }
};

// Binary quantization with Scalar4bits / Scalar8bits queries (scan_bq.hip RowBQScalar<B>): the plane-weighted xor-popcount of a row v against a
// query whose values are t_i in [0, 2^B - 1] is  sum_i (v_i ? 2^B - 1 - t_i : t_i)  =  sum t + (2^B - 1) ones(v) - 2 v . t  - the row's bits
// expanded to 0 / 1 bytes as in Tq1Ops against the values as i8 (8-bit ones less 128: + 128 ones(v)), bq_encode_scalar_query_kernel writes them and
// sum t behind the planes.  Same integer, then calculate_metric's f32 expression.
template <int B>
struct BqOps {
    typedef i32x4 acc_t;
    static constexpr int NA = 1;
    static constexpr uint32_t QSTEP = 4 * 128;
    static constexpr bool ROW_ONES = true;
    typedef typename Tq1Ops<1, false>::dec_t dec_t;
    static __device__ __forceinline__ uint32_t body_bytes(const ScanArgs &a) { return a.dim; }
    static __device__ __forceinline__ uint32_t query_off(const ScanArgs &a) { return a.tq_qbytes_off; }
    static __device__ __forceinline__ uint32_t kg_off(int kg) { return Tq1Ops<1, false>::kg_off(kg); }
    static __device__ __forceinline__ void decode(const uint4 &x, dec_t &d) { Tq1Ops<1, false>::decode(x, d); }
    static __device__ __forceinline__ void mac(const dec_t &d, const unsigned char *qp, acc_t (&acc)[NA]) { Tq1Ops<1, false>::mac(d, qp, acc); }
    static __device__ __forceinline__ float row_aux(const ScanArgs &, uint32_t) { return 0.0f; }
    struct qc_t { int32_t sum_t; };
    static __device__ __forceinline__ qc_t load_qc(const ScanArgs &a, const unsigned char *q_entry) {
        return qc_t{(int32_t)reinterpret_cast<const QueryAux *>(q_entry + a.aux_off)->pad[1]};
    }
    static __device__ __forceinline__ float finish(const ScanArgs &a, const acc_t (&acc)[NA], int r, const qc_t &qc, const unsigned char *, uint32_t,
                                                   float, uint32_t ones) {
        constexpr int32_t M = (1 << B) - 1;
        const int32_t v_dot_t = acc[0][r] + (B == 8 ? 128 * (int32_t)ones : 0);
        const uint32_t weighted = (uint32_t)(qc.sum_t + M * (int32_t)ones - 2 * v_dot_t);
        const float xor_product = (float)weighted / (float)M;   // calculate_metric (encoded_vectors_binary.rs:766-810)
        const float zeros_count = (float)a.bq_dim - xor_product;
        return a.bq_flip ? xor_product - zeros_count : zeros_count - xor_product;
    }
};

struct F16Ops {    // Metric<f16> dot / cosine: f16 products are exact in f32, f32 accumulation (order differs from the
                   // x86 leaf: within 1e-5 of it, the bar of the f16 path), scalar tail as in metric_f16/avx/dot.rs:64-66
    typedef f32x4 acc_t;
    static constexpr int NA = 1;
    static constexpr uint32_t QSTEP = 64;
    typedef uint4 dec_t;
    static __device__ __forceinline__ uint32_t body_bytes(const ScanArgs &a) { return a.tail_start * 2; }
    static __device__ __forceinline__ void decode(const uint4 &x, dec_t &d) { d = x; }
    static __device__ __forceinline__ void mac(const dec_t &d, const unsigned char *qp, acc_t (&acc)[NA]) {
        const uint4 y = *reinterpret_cast<const uint4 *>(qp);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8 *>(&d), *reinterpret_cast<const f16x8 *>(&y), acc[0], 0, 0, 0);
    }
    static __device__ __forceinline__ float row_aux(const ScanArgs &, uint32_t) { return 0.0f; }
    struct qc_t { const unsigned char *q_entry; };
    static __device__ __forceinline__ qc_t load_qc(const ScanArgs &, const unsigned char *q_entry) { return qc_t{q_entry}; }
    static __device__ __forceinline__ float finish(const ScanArgs &a, const acc_t (&accs)[NA], int r, const qc_t &qc, const unsigned char *row, uint32_t,
                                                   float, uint32_t) {
        float result = accs[0][r];
        const _Float16 *qh = reinterpret_cast<const _Float16 *>(qc.q_entry);
        const _Float16 *vh = reinterpret_cast<const _Float16 *>(row);
        for (uint32_t i = a.tail_start; i < a.dim; ++i) result += (float)qh[i] * (float)vh[i];
        return result;
    }
};

constexpr int SQM_BLOCK = 512;
constexpr int SQM_NW = SQM_BLOCK / WAVE;

// Two variants of this kernel were built in round 4, measured and are gone from the code since round 6 (profiles/r4_c3_sq_scan_staged_vs_direct.md): rows
// staged through wave-private LDS buffers as whole 1 KiB runs (0.723 against 0.751 of HBM at 32 queries: slower), and the waves' top lists in LDS behind the
// query tile instead of in registers (twice the waves per SIMD, the same time for SQ, 1.75 against 1.36 ms for the decode-bound TurboQuant scan).
template <class Ops, int QW, int D, bool HAS_IDS, int MODE>
__global__ __launch_bounds__(SQM_BLOCK) void scan_sq_mfma_kernel(const ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NG = QW / 16;
    if (a.run_if && *a.run_if == 0) return;           // the exact pass behind the 128-query TurboQuant pass was not needed (scan_tq4w.hip)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.queries);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        const uint32_t n16 = (uint32_t)QW * a.q_stride / 16;
        for (uint32_t i = tid; i < n16; i += SQM_BLOCK) dst[i] = src[i];
    }
    __syncthreads();

    const int n = lane & 15;       // A: stored row of the tile; B / D: query of the group
    const int kg = lane >> 4;      // which 16 bytes of the 64-byte step; D: rows 4 kg .. 4 kg + 3
    const unsigned char *rows = reinterpret_cast<const unsigned char *>(a.rows);
    const uint32_t nbytes = Ops::body_bytes(a);             // bytes of the SIMD body per row (multiple of 16)
    const uint32_t nstep = (nbytes + 63) / 64;
    const unsigned char *qbase = smem + (uint32_t)n * a.q_stride + ops_query_off<Ops>::get(a) + ops_kg_off<Ops>::get(kg);   // + g * 16 * q_stride + s * QSTEP
    const uint32_t gstride = 16u * a.q_stride;
    const int top = (int)a.top;

    typename Ops::qc_t qc[NG];     // the lane's query (16 g + n of the tile) as finish() needs it
#pragma unroll
    for (int g = 0; g < NG; ++g) qc[g] = Ops::load_qc(a, smem + (uint32_t)(16 * g + n) * a.q_stride);

    uint64_t list[QW];
    uint64_t thr[NG];              // reject bound of query 16 g + n: the k-th best key of the wave's list, never below ...
    uint64_t gk[NG];               // ... the score part of the pre-scan's bound (api_search.hip search_enqueue; 0 = none): equal scores pass
    float thr_f[NG];               // the score of thr (-inf without one): one float compare rejects a pair before its key is even made
#pragma unroll
    for (int q = 0; q < QW; ++q) list[q] = 0;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const uint32_t q = (uint32_t)(16 * g + n);
        gk[g] = (MODE == SCAN_TOPK && a.gthr && q < a.nq) ? (a.gthr[q] & 0xFFFFFFFF00000000ull) : 0ull;
        thr[g] = gk[g];
        thr_f[g] = gk[g] ? key_score(gk[g]) : -__builtin_inff();
    }

    const uint32_t gw = blockIdx.x * SQM_NW + wave;
    const uint32_t tw = gridDim.x * SQM_NW;
    const uint64_t n_tiles = (a.n_cand + 15) / 16;

    auto row_of = [&](uint64_t tile, int m, bool *ok) -> uint32_t {
        const uint64_t c = tile * 16 + (uint32_t)m;
        bool v = c < a.n_cand;
        uint32_t id = HAS_IDS ? a.ids[v ? c : 0] : (uint32_t)(v ? c : 0);
        if (HAS_IDS && id >= a.n_rows) {
            if (v) *a.err_flag = 1;
            id = 0;
            v = false;
        }
        if (ok) *ok = v;
        return id;
    };
    // the step's 16 bytes of this lane; bytes past the row (last, partial step) are zero and never read
    auto load_piece = [&](const unsigned char *rp, uint32_t s) -> uint4 {
        const uint32_t off = s * 64 + (uint32_t)kg * 16;
        // plain (temporal) loads: the 4 lanes of a row touch half a 128-byte line per step, the other half is the next
        // step's load — it must still be in L1 / L2 then
        if (off < nbytes) return *reinterpret_cast<const uint4 *>(rp + off);
        return make_uint4(0, 0, 0, 0);
    };

    uint4 cur[D], nxt[D];
    const unsigned char *rp = rows;
    rp = rows + (uint64_t)row_of(gw < n_tiles ? gw : 0, n, nullptr) * a.row_stride;
#pragma unroll
    for (int d = 0; d < D; ++d) cur[d] = load_piece(rp, (uint32_t)d < nstep ? d : 0);

    for (uint64_t tile = gw; tile < n_tiles; tile += tw) {
        // the 4 result rows of this lane (rows 4 kg + r) and their vector offsets
        uint32_t rid[4];
        bool valid[4];
        float v_off[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rid[r] = row_of(tile, 4 * kg + r, &valid[r]);
            v_off[r] = Ops::row_aux(a, rid[r]);
        }

        typename Ops::acc_t acc[NG][Ops::NA];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int k = 0; k < Ops::NA; ++k) acc[g][k] = (typename Ops::acc_t){0, 0, 0, 0};

        uint32_t ones = 0;
        const unsigned char *rp_next = rows + (uint64_t)row_of(tile + tw < n_tiles ? tile + tw : 0, n, nullptr) * a.row_stride;
        for (uint32_t s0 = 0; s0 < nstep; s0 += D) {
            const bool last_chunk = s0 + D >= nstep;
            const unsigned char *np = last_chunk ? rp_next : rp;
            const uint32_t ns0 = last_chunk ? 0 : s0 + D;
#pragma unroll
            for (int d = 0; d < D; ++d) nxt[d] = load_piece(np, ns0 + d < nstep ? ns0 + d : nstep - 1);
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (s0 + d < nstep) {
                    typename Ops::dec_t dec;
                    Ops::decode(cur[d], dec);
                    if (ops_row_ones<Ops>::value) ones += __popc(cur[d].x) + __popc(cur[d].y) + __popc(cur[d].z) + __popc(cur[d].w);
#pragma unroll
                    for (int g = 0; g < NG; ++g) Ops::mac(dec, qbase + (uint32_t)g * gstride + (s0 + d) * Ops::QSTEP, acc[g]);
                }
            }
#pragma unroll
            for (int d = 0; d < D; ++d) cur[d] = nxt[d];
        }
        rp = rp_next;
        uint32_t row_ones[4] = {0, 0, 0, 0};
        if (ops_row_ones<Ops>::value) {   // lanes n, n + 16, n + 32, n + 48 hold row n's pieces; the results of rows 4 kg + r sit in this lane
            ones += __shfl_xor(ones, 16);
            ones += __shfl_xor(ones, 32);
#pragma unroll
            for (int r = 0; r < 4; ++r) row_ones[r] = __shfl(ones, 4 * kg + r);
        }

        // ---- postprocess_score (multiplier * dot + query_offset + vector_offset, left to right, not fused) + top-k ----
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const uint32_t q = (uint32_t)(16 * g + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float score = Ops::finish(a, acc[g], r, qc[g], rows + (uint64_t)rid[r] * a.row_stride, rid[r], v_off[r], row_ones[r]);
                const bool mine = valid[r] && q < a.nq;
                if (MODE == SCAN_SCORES) {
                    if (mine) a.scores[(uint64_t)q * a.scores_stride + (tile * 16 + (uint32_t)(4 * kg + r))] = score;
                } else {
                    // cheap reject on the score alone (the epilogue is a third of the kernel's issue slots at 32 queries); ties with the k-th score
                    // and NaN (greatest in OrderedFloat) fall through to the exact key compare
                    bool c = mine && !(score < thr_f[g]);
                    if (__ballot(c)) {
                        const uint64_t key = make_key(score, rid[r]);
                        c = c && key > thr[g] && a.del.live(rid[r]) && (!a.key_bound || key < a.key_bound[q]);
                        uint64_t mask = __ballot(c);
                        while (mask) {
                            const int src = __builtin_ctzll(mask);
                            mask &= mask - 1;
                            const uint64_t nk = readlane_u64(key, src);
                            const int ql = 16 * g + (src & 15);
#pragma unroll
                            for (int qq = 16 * g; qq < 16 * g + 16; ++qq) {
                                if (ql == qq) {
                                    if (nk > readlane_u64(list[qq], top - 1)) {
                                        wave_list_insert(list[qq], nk, lane);
                                        const uint64_t nt = readlane_u64(list[qq], top - 1);
                                        if (n == qq - 16 * g) {
                                            thr[g] = nt > gk[g] ? nt : gk[g];
                                            thr_f[g] = thr[g] ? key_score(thr[g]) : -__builtin_inff();
                                        }
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

    // ---- block merge: 8 wave lists -> 1 list per query, one global write per block ----
    __syncthreads();
    uint64_t *lds_keys = reinterpret_cast<uint64_t *>(smem);
    const uint32_t utop = a.top;
#pragma unroll
    for (int q = 0; q < QW; ++q)
        if (lane < top) lds_keys[((uint32_t)wave * QW + q) * utop + lane] = list[q];
    __syncthreads();
    for (uint32_t q = wave; q < a.nq; q += SQM_NW) {
        uint64_t merged = 0;
        for (int sw = 0; sw < SQM_NW; ++sw) {
            const uint64_t key = lane < top ? lds_keys[((uint32_t)sw * QW + q) * utop + lane] : 0;
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

template <class Ops, int QW, int D, bool HAS_IDS, int MODE>
static int32_t launch_sqm_inst(hipStream_t st, const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    size_t lds = ((size_t)QW * a.q_stride + 15) & ~(size_t)15;
    if (MODE == SCAN_TOPK) {
        const size_t lk = (size_t)SQM_NW * QW * a.top * sizeof(uint64_t);
        if (lk > lds) lds = lk;            // the waves' lists are parked over the query tile for the block merge
    }
    lds = (lds + 15) & ~(size_t)15;
    QMX_REQUIRE(lds <= 160 * 1024, QMX_ERR_NOT_SUPPORTED, "query tile needs %zu B of LDS (> 160 KiB)", lds);
    auto kfn = scan_sq_mfma_kernel<Ops, QW, D, HAS_IDS, MODE>;
    static thread_local DeviceOnce attr_once;
    if (attr_once.need()) {
        QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_once.mark();
    }
    int per_cu = 0;
    QMX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, SQM_BLOCK, lds));
    if (per_cu < 1) per_cu = 1;
    const uint64_t n_tiles = (a.n_cand + 15) / 16;
    const uint64_t want = (n_tiles + SQM_NW - 1) / SQM_NW;
    const uint64_t cap = (uint64_t)num_cus * per_cu;
    uint32_t grid = (uint32_t)(want < cap ? want : cap);
    if (grid < 1) grid = 1;
    if (grid_out) {
        if (*grid_out && MODE == SCAN_TOPK && grid > *grid_out) grid = *grid_out;
        *grid_out = grid;
    }
    ::qmx::clear_stale_error();
    QMX_NOTE_KERNEL(kfn);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(SQM_BLOCK), lds, st, a);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

template <class Ops, int QW, int D>
static int32_t launch_sqm_qt(hipStream_t st, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid) {
    const bool ids = a.ids != nullptr;
    if (mode == SCAN_TOPK)
        return ids ? launch_sqm_inst<Ops, QW, D, true, SCAN_TOPK>(st, a, num_cus, grid) : launch_sqm_inst<Ops, QW, D, false, SCAN_TOPK>(st, a, num_cus, grid);
    return ids ? launch_sqm_inst<Ops, QW, D, true, SCAN_SCORES>(st, a, num_cus, grid) : launch_sqm_inst<Ops, QW, D, false, SCAN_SCORES>(st, a, num_cus, grid);
}

// the f32 adds of the AVX2 leaf stay exact (and its result order-free) while every partial sum is < 2^24
bool sq_mfma_ok(uint32_t distance, uint32_t actual_dim) {
    return distance != QMX_DISTANCE_MANHATTAN && (uint64_t)127 * 127 * actual_dim < (1ull << 24);
}

// qt in {8, 16, 32}: 8 and 16 run the 16-query kernel (one accumulator), 32 two accumulators
int32_t launch_scan_sq_mfma(hipStream_t st, int qt, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    switch (qt) {
        case 8:
        case 16: return launch_sqm_qt<SqOps, 16, 6>(st, mode, a, num_cus, grid_out);
        case 32: return launch_sqm_qt<SqOps, 32, 6>(st, mode, a, num_cus, grid_out);
    }
    set_error("unsupported SQ MFMA query tile %d", qt);
    return QMX_ERR_BAD_ARG;
}

// BQ rows against Scalar4bits / Scalar8bits queries, 4..32 queries per pass
int32_t launch_scan_bq_mfma(hipStream_t st, int qt, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    if (a.bq_qbits == 4) return qt <= 16 ? launch_sqm_qt<BqOps<4>, 16, 2>(st, mode, a, num_cus, grid_out) : launch_sqm_qt<BqOps<4>, 32, 2>(st, mode, a, num_cus, grid_out);
    if (a.bq_qbits == 8) return qt <= 16 ? launch_sqm_qt<BqOps<8>, 16, 2>(st, mode, a, num_cus, grid_out) : launch_sqm_qt<BqOps<8>, 32, 2>(st, mode, a, num_cus, grid_out);
    set_error("BQ matrix-core scan: %u-bit queries not supported", a.bq_qbits);
    return QMX_ERR_NOT_SUPPORTED;
}

// TurboQuant, 4..32 queries per pass
int32_t launch_scan_tq_mfma(hipStream_t st, int qt, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    const bool l2 = a.tq_l2 != nullptr;
#define QMX_TQM(B, L)                                                                       \
    if (a.tq_bits == B && l2 == L) {                                                        \
        if (qt <= 16) return launch_sqm_qt<TqOps<B, L>, 16, 4>(st, mode, a, num_cus, grid_out); \
        return launch_sqm_qt<TqOps<B, L>, 32, 4>(st, mode, a, num_cus, grid_out);           \
    }
    QMX_TQM(4, false)
    QMX_TQM(4, true)
    QMX_TQM(2, false)
    QMX_TQM(2, true)
#undef QMX_TQM
    if (a.tq_bits == 1) {   // 8-bit query values: one accumulator; 16-bit (TQ+): two halves
#define QMX_TQ1(NA, L)                                                                         \
        if ((a.tq_planes == 16 ? 2 : 1) == NA && l2 == L) {                                      \
            if (qt <= 16) return launch_sqm_qt<Tq1Ops<NA, L>, 16, 2>(st, mode, a, num_cus, grid_out); \
            return launch_sqm_qt<Tq1Ops<NA, L>, 32, 2>(st, mode, a, num_cus, grid_out);           \
        }
        QMX_TQ1(1, false)
        QMX_TQ1(1, true)
        QMX_TQ1(2, false)
        QMX_TQ1(2, true)
#undef QMX_TQ1
    }
    set_error("TurboQuant matrix-core scan: %u bits not supported", a.tq_bits);
    return QMX_ERR_NOT_SUPPORTED;
}

// f16 rows, dot / cosine, dim >= 32: v_mfma_f32_16x16x32_f16, same streaming structure (64 row bytes = 32 halfs per step)
int32_t launch_scan_f16_mfma(hipStream_t st, int qt, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    switch (qt) {
        case 8:
        case 16: return launch_sqm_qt<F16Ops, 16, 6>(st, mode, a, num_cus, grid_out);
        case 32: return launch_sqm_qt<F16Ops, 32, 6>(st, mode, a, num_cus, grid_out);
    }
    set_error("unsupported f16 MFMA query tile %d", qt);
    return QMX_ERR_BAD_ARG;
}

}  // namespace qmx

// hnsw_build_tq.hip - the HNSW build through the TurboQuant scorer (hnsw_build.hpp).
#include "tq_policies.hpp"

namespace qmx {

// ------------------------------------------------------------------------------------------
// HNSW build over a TurboQuant segment (hnsw/build.rs:334-341 + point_scorer.rs:183-218).  EncodedVectorsTQ cannot turn a stored row into a
// query (encode_internal_vector -> None), so - as for PQ - the searches of an insertion score through precompute_query of the point's ORIGINAL
// vector (the RowTQ* policies over the batch's entries, made by api_hnsw.hip before phase 1 and staged in LDS per insertion) while everything
// stored <-> stored - the heuristic, the back links, an entry point at or below the new point's level - is score_symmetric
// (turboquant/quantization.rs:395-494): the integer dot of the two rows' codebook bytes (1 bit: dim - 2 popcount(a ^ b)), as tq_internal_kernel
// below.  One lane per stored row; `qp` is the other row's code bytes inside the block, so its index - the extras columns - follows from the pointer.
// ------------------------------------------------------------------------------------------
__device__ __constant__ int8_t TQ4_SIGNED_B[16] = {-128, -97, -76, -59, -44, -31, -18, -6, 6, 18, 31, 44, 59, 76, 97, 127};
__device__ __constant__ int8_t TQ2_SIGNED_B[4] = {-128, -38, 38, 127};
template <int BITS, bool L2>
struct HopTQInternal {
    static constexpr int LPI = 1;
    static constexpr bool MULTI = false;
    static constexpr bool INTERNAL_QOFF = false;
    static constexpr bool INTERNAL_NORM = false;
    static constexpr bool ASYMMETRIC = true;
    static __device__ __forceinline__ float score(const ScanArgs &a, const unsigned char *qp, uint32_t id, int) {
        const unsigned char *rows = reinterpret_cast<const unsigned char *>(a.rows);
        const uint32_t ia = (uint32_t)((uint64_t)(qp - rows) / a.row_stride), ib = id;
        const unsigned char *ra = qp, *rb = rows + (uint64_t)ib * a.row_stride;
        const uint32_t nb = a.tq_code_bytes, nd = nb / 4;
        const uint32_t *wa = reinterpret_cast<const uint32_t *>(ra), *wb = reinterpret_cast<const uint32_t *>(rb);
        float raw_dot;
        if (BITS == 1) {
            uint32_t pop = 0;
            for (uint32_t w = 0; w < (nb + 3) / 4; ++w) pop += (uint32_t)__popc(wa[w] ^ wb[w]);   // the block's padding bytes are zero in both rows
            const int64_t sign_sum = (int64_t)nb * 8 - 2 * (int64_t)pop;
            const float centroid_sq = 0.7978846f * 0.7978846f;
            raw_dot = centroid_sq * (float)sign_sum;
        } else if (a.tq_ec.weights) {   // score_symmetric_ec: the i16 weight of every coordinate
            const int16_t *wt = a.tq_ec.weights;
            int64_t acc = 0;
            if (BITS == 4) {
                for (uint32_t k = 0; k < nb; ++k)
                    acc += (int64_t)TQ4_SIGNED_B[ra[k] & 15] * TQ4_SIGNED_B[rb[k] & 15] * wt[2 * k] + (int64_t)TQ4_SIGNED_B[ra[k] >> 4] * TQ4_SIGNED_B[rb[k] >> 4] * wt[2 * k + 1];
            } else {
                for (uint32_t k = 0; k < nb; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc += (int64_t)TQ2_SIGNED_B[(ra[k] >> (2 * j)) & 3] * TQ2_SIGNED_B[(rb[k] >> (2 * j)) & 3] * wt[4 * k + j];
            }
            const float codebook_scale = 128.0f / (BITS == 4 ? 2.733f : 1.510f);
            const float weighted = (float)acc / (a.tq_ec.weight_scale * (codebook_scale * codebook_scale));
            raw_dot = ((weighted + a.tq_ec.xm[ia]) + a.tq_ec.xm[ib]) - a.tq_ec.mm_const;
        } else {
            int32_t acc = 0;   // |c_a c_b| <= 2^14 per coordinate: exact in i32 below 2^17 coordinates
            if (BITS == 4) {
                for (uint32_t w = 0; w < nd; ++w) {
                    const uint32_t x = wa[w], y = wb[w];
                    acc = sdot4(tq4_lookup(x & 0x0F0F0F0Fu), tq4_lookup(y & 0x0F0F0F0Fu), acc);
                    acc = sdot4(tq4_lookup((x >> 4) & 0x0F0F0F0Fu), tq4_lookup((y >> 4) & 0x0F0F0F0Fu), acc);
                }
                for (uint32_t k = nd * 4; k < nb; ++k)
                    acc += (int32_t)TQ4_SIGNED_B[ra[k] & 15] * TQ4_SIGNED_B[rb[k] & 15] + (int32_t)TQ4_SIGNED_B[ra[k] >> 4] * TQ4_SIGNED_B[rb[k] >> 4];
            } else {
                for (uint32_t w = 0; w < nd; ++w) {
                    const uint32_t x = wa[w], y = wb[w];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc = sdot4(__builtin_amdgcn_perm(0u, TQ2_T, (x >> (2 * j)) & 0x03030303u), __builtin_amdgcn_perm(0u, TQ2_T, (y >> (2 * j)) & 0x03030303u), acc);
                }
                for (uint32_t k = nd * 4; k < nb; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc += (int32_t)TQ2_SIGNED_B[(ra[k] >> (2 * j)) & 3] * TQ2_SIGNED_B[(rb[k] >> (2 * j)) & 3];
            }
            const float codebook_scale = 128.0f / (BITS == 4 ? 2.733f : 1.510f);
            raw_dot = (float)acc / (codebook_scale * codebook_scale);
        }
        const float s1 = a.tq_sf[ia], s2 = a.tq_sf[ib];
        float score;
        if (L2) {
            const float x = a.tq_l2[ia], y = a.tq_l2[ib];
            score = (x * x + y * y) - ((2.0f * s1) * s2) * raw_dot;
        } else {
            score = (raw_dot * s1) * s2;
        }
        return a.tq_invert ? -score : score;
    }
};
int32_t launch_hnsw_build_tq(hipStream_t st, const ScanArgs &a, const HnswBuildArgs &h, int phase, uint32_t grid, int *per_cu) {
    QMX_REQUIRE(h.batch_queries, QMX_ERR_BAD_ARG, "TurboQuant build needs the batch's query entries");
    const bool l2 = a.tq_l2 != nullptr;
#define QMX_TQB(B, L, ROW)                                                                                                      \
    if (a.tq_bits == B && l2 == L) return launch_hnsw_build_hop<HopRow<ROW>, HopTQInternal<B, L>>(st, a, h, phase, grid, per_cu);
    QMX_TQB(4, false, RowTQ4<false>)
    QMX_TQB(4, true, RowTQ4<true>)
    QMX_TQB(2, false, RowTQ2<false>)
    QMX_TQB(2, true, RowTQ2<true>)
#undef QMX_TQB
    if (a.tq_bits == 1) {
        if (a.tq_planes == 16)
            return l2 ? launch_hnsw_build_hop<HopRow<RowTQ1<true, 16>>, HopTQInternal<1, true>>(st, a, h, phase, grid, per_cu)
                      : launch_hnsw_build_hop<HopRow<RowTQ1<false, 16>>, HopTQInternal<1, false>>(st, a, h, phase, grid, per_cu);
        return l2 ? launch_hnsw_build_hop<HopRow<RowTQ1<true>>, HopTQInternal<1, true>>(st, a, h, phase, grid, per_cu)
                  : launch_hnsw_build_hop<HopRow<RowTQ1<false>>, HopTQInternal<1, false>>(st, a, h, phase, grid, per_cu);
    }
    set_error("TurboQuant build: %u bits per value not supported", a.tq_bits);
    return QMX_ERR_NOT_SUPPORTED;
}

}  // namespace qmx

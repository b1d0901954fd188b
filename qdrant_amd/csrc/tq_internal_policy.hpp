// tq_internal_policy.hpp - score_symmetric of two TurboQuant rows as a hop scorer (the stored <-> stored scores of an HNSW build: hnsw_build_tq.hip for
// single vectors, hnsw_build_multi_tq.hip under MaxSim for multi-vector points).
#pragma once
#include "tq_policies.hpp"

namespace qmx {

static __device__ __constant__ int8_t TQ4_SIGNED_B[16] = {-128, -97, -76, -59, -44, -31, -18, -6, 6, 18, 31, 44, 59, 76, 97, 127};
static __device__ __constant__ int8_t TQ2_SIGNED_B[4] = {-128, -38, 38, 127};
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
}  // namespace qmx

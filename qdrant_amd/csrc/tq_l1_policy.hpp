// tq_l1_policy.hpp - the hop scorer of a walk THROUGH an EncodedVectorsTQ storage over Distance::Manhattan (DistanceType::L1): every hop candidate is
// dequantised, rotated back and compared with the query as given (TurboQuantizer::score_precomputed, turboquant/quantization.rs:596-607; tq_l1.hip has
// the batch form of the same arithmetic).  A wave works on one candidate at a time - the row in registers, element i in lane i / E, the inverse rotation
// of tq_rotate.hpp - and parks the |q - v| terms of up to G candidates in LDS; then G lanes add their candidate's terms in index order (the reference's
// f32 iterator sum: one chain per score).  Slow next to the integer scorers, as in the reference - and what a Manhattan collection with TurboQuant walks with.
#pragma once
#include "hnsw.hpp"
#include "tq_rotate.hpp"

namespace qmx {

// element i of `row`, dequantised in rotated space (dequantize, quantization.rs:321-376): the centroid, the TQ+ correction reverted, times
// recovered_l2 / sqrt(padded_dim) with recovered_l2 = the row's scaling factor
__device__ __forceinline__ double tq_l1_dequant_elem(const TqL1Dev &d, const unsigned char *row, double l1_scale, uint32_t i) {
    const float C1[2] = {-0.7978846f, 0.7978846f};
    const float C2[4] = {-1.510f, -0.4528f, 0.4528f, 1.510f};
    const float C4[16] = {-2.733f, -2.069f, -1.618f, -1.256f, -0.9424f, -0.6568f, -0.3881f, -0.1284f, 0.1284f, 0.3881f, 0.6568f, 0.9424f, 1.256f, 1.618f, 2.069f, 2.733f};
    const uint32_t vb = d.value_bits, bit = i * vb;
    const uint32_t code = (row[bit >> 3] >> (bit & 7u)) & ((1u << vb) - 1u);
    double c = (double)(vb == 4 ? C4[code] : vb == 2 ? C2[code & 3u] : C1[code & 1u]);
    if (d.shift) c = c / (double)d.scale[i] - (double)d.shift[i];
    return c * l1_scale;
}

// score_symmetric's L1 arm as a hop scorer (quantization.rs:429-440; the heuristic, the back links and an entry point at or below the new point's level of
// an HNSW build through such a storage, hnsw_build_tq_l1.hip): both rows dequantised, ONE inverse rotation of their difference, the f32 sum of |x| over all
// padded_dim coordinates in index order.  `qp` = the code bytes of the stored row that plays the query.  The coordinates past the rotation (an unpadded
// rotation of a 1.5-bit code) are not rotated: their terms are taken before the transform, which scrambles the registers of the lanes that hold them.
template <int E>
struct HopTQL1Internal {
    static constexpr int LPI = 1;
    static constexpr bool INTERNAL_QOFF = false;
    static constexpr bool INTERNAL_NORM = false;
    static constexpr bool MULTI = false;
    static constexpr bool TQL1 = true;
    static constexpr bool ASYMMETRIC = true;
    static constexpr uint32_t G = E == 16 ? 8 : E == 32 ? 4 : 1;         // candidates whose terms are parked together (~32 KiB)
    static __device__ __forceinline__ float score(const ScanArgs &, const unsigned char *, uint32_t, int) { return 0.0f; }   // (never called: hop() scores)

    static __device__ __forceinline__ void hop(const ScanArgs &a, const unsigned char *qp, const uint32_t *hop_ids, float *hop_scores, uint32_t k, int lane) {
        __shared__ double buf[64 * E];
        __shared__ float terms[G * (64 * E + 1)];
        const TqL1Dev &d = *reinterpret_cast<const TqL1Dev *>(a.tq_l1);
        const TqRotation r = d.inv;
        const uint32_t pd = r.padded_dim, tstride = pd + 1u;
        const uint32_t first = (uint32_t)lane * E;
        const bool act = first < r.rot_dim;
        uint32_t my_size;
        double my_norm;
        tq_wave_lane_chunk<E>(r, lane, &my_size, &my_norm);
        const double sqrt_pd = sqrt((double)pd);
        const unsigned char *rows = reinterpret_cast<const unsigned char *>(a.rows);
        const uint32_t ia = (uint32_t)((uint64_t)(qp - rows) / a.row_stride);
        const double scale_a = (double)a.tq_sf[ia] / sqrt_pd;
        for (uint32_t j0 = 0; j0 < k; j0 += G) {
            const uint32_t g = k - j0 < G ? k - j0 : G;
            for (uint32_t jj = 0; jj < g; ++jj) {
                const uint32_t id = hop_ids[j0 + jj];
                const unsigned char *rb = rows + (uint64_t)id * a.row_stride;
                const double scale_b = (double)a.tq_sf[id] / sqrt_pd;
                double x[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const uint32_t i = first + (uint32_t)e;
                    x[e] = i < pd ? tq_l1_dequant_elem(d, qp, scale_a, i) - tq_l1_dequant_elem(d, rb, scale_b, i) : 0.0;
                }
                if (!act) {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const uint32_t i = first + (uint32_t)e;
                        if (i < pd) terms[jj * tstride + i] = (float)__builtin_fabs(x[e]);
                    }
                }
                tq_wave_rotate<E>(x, r, buf, my_size, my_norm, lane);
                if (act) {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const uint32_t i = first + (uint32_t)e;
                        if (i < pd) terms[jj * tstride + i] = (float)__builtin_fabs(x[e]);
                    }
                }
            }
            __syncthreads();
            if ((uint32_t)lane < g) {
                float sum = 0.0f;
                for (uint32_t i = 0; i < pd; ++i) sum = sum + terms[(uint32_t)lane * tstride + i];
                hop_scores[j0 + (uint32_t)lane] = a.tq_invert ? -sum : sum;
            }
            __syncthreads();
        }
    }
};

template <int E>
struct HopTQL1 {
    static constexpr int LPI = 1;
    static constexpr bool INTERNAL_QOFF = false;
    static constexpr bool INTERNAL_NORM = false;
    static constexpr bool MULTI = false;
    static constexpr bool TQL1 = true;
    static __device__ __forceinline__ float score(const ScanArgs &, const unsigned char *, uint32_t, int) { return 0.0f; }   // (never called: hop() scores)

    static __device__ __forceinline__ uint32_t scratch_bytes(const ScanArgs &a) {
        const TqL1Dev &d = *reinterpret_cast<const TqL1Dev *>(a.tq_l1);
        return tq_l1_lds_bytes(d.dim, d.inv.rot_dim) - tq_l1_query_bytes(d.dim);
    }
    // the hop scratch lies behind the query entry (tq_l1_lds_bytes) ...
    static __device__ __forceinline__ void hop(const ScanArgs &a, const unsigned char *qp, const uint32_t *hop_ids, float *hop_scores, uint32_t k, int lane) {
        const TqL1Dev &d = *reinterpret_cast<const TqL1Dev *>(a.tq_l1);
        hop_at(a, qp, const_cast<unsigned char *>(qp) + tq_l1_query_bytes(d.dim), hop_ids, hop_scores, k, lane);
    }
    // ... or where the caller keeps it (HopCustom: behind the examples' entries, one scratch for all of them): tq_l1_lds_bytes - tq_l1_query_bytes bytes of LDS
    static __device__ __forceinline__ void hop_at(const ScanArgs &a, const unsigned char *qp, unsigned char *scratch, const uint32_t *hop_ids, float *hop_scores,
                                                  uint32_t k, int lane) {
        const TqL1Dev &d = *reinterpret_cast<const TqL1Dev *>(a.tq_l1);
        const TqRotation r = d.inv;
        const uint32_t dim = d.dim;
        const float *q = reinterpret_cast<const float *>(qp);
        double *buf = reinterpret_cast<double *>(scratch);
        float *terms = reinterpret_cast<float *>(buf + r.rot_dim);
        const uint32_t G = tq_l1_group(dim), tstride = dim + 1u;
        const uint32_t first = (uint32_t)lane * E;
        const bool act = first < r.rot_dim;
        uint32_t my_size;
        double my_norm;
        tq_wave_lane_chunk<E>(r, lane, &my_size, &my_norm);
        const double sqrt_pd = sqrt((double)r.padded_dim);
        const unsigned char *rows = reinterpret_cast<const unsigned char *>(a.rows);
        for (uint32_t j0 = 0; j0 < k; j0 += G) {
            const uint32_t g = k - j0 < G ? k - j0 : G;
            for (uint32_t jj = 0; jj < g; ++jj) {
                const uint32_t id = hop_ids[j0 + jj];
                const unsigned char *row = rows + (uint64_t)id * a.row_stride;
                const double l1_scale = (double)a.tq_sf[id] / sqrt_pd;
                double x[E];
#pragma unroll
                for (int e = 0; e < E; ++e) x[e] = act ? tq_l1_dequant_elem(d, row, l1_scale, first + (uint32_t)e) : 0.0;
                tq_wave_rotate<E>(x, r, buf, my_size, my_norm, lane);
                if (act) {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const uint32_t i = first + (uint32_t)e;
                        if (i < dim) terms[jj * tstride + i] = (float)__builtin_fabs((double)q[i] - x[e]);
                    }
                }
            }
            __syncthreads();
            if ((uint32_t)lane < g) {
                float sum = 0.0f;
                for (uint32_t i = 0; i < dim; ++i) sum = sum + terms[(uint32_t)lane * tstride + i];
                hop_scores[j0 + (uint32_t)lane] = a.tq_invert ? -sum : sum;
            }
            __syncthreads();
        }
    }
};

}  // namespace qmx

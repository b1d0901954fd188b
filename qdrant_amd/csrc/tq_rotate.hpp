// tq_rotate.hpp - HadamardRotation on the device, the pieces shared by the encoders (scan_tq.hip) and the walk through a TurboQuant storage over
// Manhattan (tq_l1_policy.hpp): the tables of a rotation and the one-vector-per-wave transform.
#pragma once
#include "common.hpp"

namespace qmx {

struct TqRotation {
    const uint32_t *maps;       // [3][rot_dim] maps of the three gathers, in application order (forward maps; apply_inverse: the backward maps, last first)
    const uint32_t *chunk_off;  // [n_chunks] first element of each power-of-two chunk
    const uint32_t *chunk_size; // [n_chunks]
    const double *chunk_norm;   // [n_chunks] 1 / sqrt(size), computed on the host like the reference does
    uint32_t n_chunks, rot_dim, padded_dim, dim;
};

// which chunk lane `lane` (coordinates lane * E ...) lies in: its size and norm (size 0: the lane holds nothing)
template <int E>
__device__ __forceinline__ void tq_wave_lane_chunk(const TqRotation &r, int lane, uint32_t *my_size, double *my_norm) {
    const uint32_t first = (uint32_t)lane * E;
    *my_size = 0;
    *my_norm = 1.0;
    for (uint32_t c = 0; c < r.n_chunks; ++c) {
        const uint32_t off = r.chunk_off[c], size = r.chunk_size[c];
        if (first < r.rot_dim && first >= off && first < off + size) { *my_size = size; *my_norm = r.chunk_norm[c]; }
    }
}

// wht_and_gather_rounds (rotation.rs:97-129) on ONE vector held by a wave: element i in lane i / E, register i % E (rot_dim a multiple of E, <= 64 E).
// Strides below E run inside the lane, a stride h >= E pairs lane l with l ^ (h / E) while the lane's chunk is longer than h; then the chunk's norm;
// the gathers go through `buf` (rot_dim doubles of LDS owned by this wave).  The reference's adds and subtracts per element, in its order.
template <int E>
__device__ __forceinline__ void tq_wave_rotate(double (&x)[E], const TqRotation &r, double *buf, uint32_t my_size, double my_norm, int lane) {
    const uint32_t first = (uint32_t)lane * E;
    const bool act = first < r.rot_dim;
    auto wht = [&]() {
#pragma unroll
        for (int h = 1; h < E; h *= 2) {
#pragma unroll
            for (int j = 0; j < E; ++j)
                if ((j & h) == 0) {
                    const double a = x[j], b = x[j + h];
                    x[j] = a + b;
                    x[j + h] = a - b;
                }
        }
        for (uint32_t hl = 1; hl < 64; hl *= 2) {
            const bool on = my_size > hl * E;
            if (!__ballot(on)) break;                          // (chunk sizes only shrink along the vector: nobody joins a later stage either)
            const bool upper = ((uint32_t)lane & hl) != 0;
#pragma unroll
            for (int k = 0; k < E; ++k) {
                const double p = __shfl_xor(x[k], (int)hl, 64);
                const double lo = upper ? p : x[k], hi = upper ? x[k] : p;      // the pair (x[j], x[j + h]) as the reference names it
                if (on) x[k] = upper ? lo - hi : lo + hi;
            }
        }
#pragma unroll
        for (int k = 0; k < E; ++k) x[k] = x[k] * my_norm;
    };
    wht();
    for (int p = 0; p < 3; ++p) {
        const uint32_t *map = r.maps + (size_t)p * r.rot_dim;
        __syncthreads();
        if (act) {
#pragma unroll
            for (int k = 0; k < E; ++k) buf[first + k] = x[k];
        }
        __syncthreads();
        if (act) {
#pragma unroll
            for (int k4 = 0; k4 < E; k4 += 4) {
                const uint4 m4 = *reinterpret_cast<const uint4 *>(map + first + k4);
                x[k4] = buf[m4.x]; x[k4 + 1] = buf[m4.y]; x[k4 + 2] = buf[m4.z]; x[k4 + 3] = buf[m4.w];
            }
        }
        wht();
    }
}

// everything the L1 scores of a TurboQuant storage need on the device (tq_l1.hip, tq_l1_policy.hpp); one per Manhattan TQ segment
struct TqL1Dev {
    TqRotation inv;             // apply_inverse's tables
    const float *shift, *scale; // TQ+ error correction, or null
    uint32_t value_bits, dim;
};

// ---- the walk through such a storage (tq_l1_policy.hpp) ----
// candidates whose terms are parked together: 48 KiB of LDS worth, 16 at most
__host__ __device__ static inline uint32_t tq_l1_group(uint32_t dim) {
    const uint32_t g = 49152u / ((dim + 1u) * 4u);
    return g > 16u ? 16u : (g < 1u ? 1u : g);
}
// LDS behind the hop buffers: [the query, f32, padded to 16 bytes][rot_dim doubles of the gathers][G x (dim + 1) terms]
__host__ __device__ static inline uint32_t tq_l1_query_bytes(uint32_t dim) { return (dim * 4u + 15u) & ~15u; }
__host__ __device__ static inline uint32_t tq_l1_lds_bytes(uint32_t dim, uint32_t rot_dim) {
    return tq_l1_query_bytes(dim) + rot_dim * 8u + ((tq_l1_group(dim) * (dim + 1u) * 4u + 15u) & ~15u);
}

}  // namespace qmx

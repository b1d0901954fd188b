// tq_l1.hip — EncodedVectorsTQ over Distance::Manhattan (DistanceType::L1, quantized_vectors.rs:230).
//
// The Hadamard rotation does not preserve L1, so the reference has no integer kernel for it: TurboQuantizer::score_precomputed (turboquant/
// quantization.rs:596-607) dequantises the stored row (dequantize :321-376: centroid values x scaling_factor / sqrt(padded_dim), the TQ+
// correction reverted), rotates it back (HadamardRotation::apply_inverse, rotation.rs:76-79) and sums |q - v| over the query's coordinates as f32,
// in order; score_symmetric (:429-440) does the same with the difference of two dequantised rows and no query.  EncodedVectorsTQ negates the sum
// (`invert`, encoded_vectors_tq.rs:510-514).  Here, per batch of rows: tq_l1_dequant_kernel -> the rotation kernel of scan_tq.hip over the backward
// maps -> tq_l1_score_kernel, one (row, query) chain per lane (the f32 sum is an iterator sum: one add per coordinate, in order).
#include "scan_common.hpp"

namespace qmx {

// out[r][i] = dequantize(row ids[r])[i], rotated space.  One thread per element.
__global__ __launch_bounds__(256) void tq_l1_dequant_kernel(const uint8_t *codes, uint64_t row_stride, const float *sf, const uint32_t *ids, uint64_t id0,
                                                            uint64_t n, uint64_t n_rows, uint32_t padded, uint32_t value_bits, const float *shift,
                                                            const float *scale, double *out, int *err_flag, PairSel sel, int use_sel) {
    const float C1[2] = {-0.7978846f, 0.7978846f};
    const float C2[4] = {-1.510f, -0.4528f, 0.4528f, 1.510f};
    const float C4[16] = {-2.733f, -2.069f, -1.618f, -1.256f, -0.9424f, -0.6568f, -0.3881f, -0.1284f, 0.1284f, 0.3881f, 0.6568f, 0.9424f, 1.256f, 1.618f, 2.069f, 2.733f};
    __shared__ float cent[16];
    if (threadIdx.x < 16) cent[threadIdx.x] = value_bits == 4 ? C4[threadIdx.x] : value_bits == 2 ? C2[threadIdx.x & 3] : C1[threadIdx.x & 1];
    __syncthreads();
    const double sqrt_pd = sqrt((double)padded);
    const uint64_t total = n * padded;
    for (uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (uint64_t)gridDim.x * 256) {
        const uint64_t r = gid / padded;
        const uint32_t i = (uint32_t)(gid % padded);
        if (use_sel && !sel.live(id0 + r, sel.query_of(id0 + r))) {      // a dead slot of a per-query list: its id is not even read
            out[gid] = 0.0;
            continue;
        }
        const uint64_t id = ids ? ids[r] : id0 + r;
        if (id >= n_rows) {
            *err_flag = 1;
            out[gid] = 0.0;
            continue;
        }
        const uint32_t bit = i * value_bits;
        const uint32_t code = (codes[id * row_stride + bit / 8] >> (bit % 8)) & ((1u << value_bits) - 1u);
        double x = (double)cent[code];
        if (shift) x = x / (double)scale[i] - (double)shift[i];           // TQ+: revert the error correction (:366-373)
        const double l1_scale = (double)sf[id] / sqrt_pd;              // recovered_l2 / sqrt(padded_dim), recovered_l2 = scaling_factor (:358)
        out[gid] = x * l1_scale;
    }
}
int32_t launch_tq_l1_dequant(hipStream_t st, const void *codes, uint64_t row_stride, const float *sf, const uint32_t *d_ids, uint64_t id0, uint64_t n,
                             uint64_t n_rows, uint32_t padded_dim, uint32_t value_bits, const float *d_shift, const float *d_scale, double *d_out, int *err_flag,
                             const PairSel *sel) {
    if (n == 0) return QMX_OK;
    const uint64_t total = n * padded_dim;
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(tq_l1_dequant_kernel, dim3((uint32_t)std::min<uint64_t>((total + 255) / 256, 1u << 20)), dim3(256), 0, st, (const uint8_t *)codes, row_stride, sf,
                       d_ids, id0, n, n_rows, padded_dim, value_bits, d_shift, d_scale, d_out, err_flag, sel ? *sel : PairSel{nullptr, 0, nullptr}, sel ? 1 : 0);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

__global__ __launch_bounds__(256) void tq_l1_diff_kernel(double *a, const double *b, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) a[i] = a[i] - b[i];
}
int32_t launch_tq_l1_diff(hipStream_t st, double *d_a, const double *d_b, uint64_t n_elems) {
    if (n_elems == 0) return QMX_OK;
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(tq_l1_diff_kernel, dim3((uint32_t)std::min<uint64_t>((n_elems + 255) / 256, 1u << 20)), dim3(256), 0, st, d_a, d_b, n_elems);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

// One wave = 64 rows of `deq`, one row per lane; 16 coordinates per trip, transposed through LDS (as tq_quantize_kernel).  QT queries per pass
// (the same for all lanes), or - PAIRS - one query per lane (items of a PairSel).
constexpr int TL1_CH = 16;
template <int QT, bool PAIRS>
__global__ __launch_bounds__(64) void tq_l1_score_kernel(const double *deq, uint64_t n, uint32_t padded, uint32_t dim, const float *queries, uint32_t q_dim,
                                                         uint32_t q0, uint32_t nq, float *scores, uint64_t stride, uint64_t col0, int invert, PairSel sel,
                                                         uint64_t item0) {
    __shared__ double tile[64][TL1_CH + 1];
    const int lane = threadIdx.x;
    const uint64_t v0 = (uint64_t)blockIdx.x * 64;
    const bool live_row = v0 + lane < n;
    uint32_t my_q = 0;
    bool live = live_row;
    if (PAIRS && live_row) {
        my_q = sel.query_of(item0 + v0 + lane);
        live = sel.live(item0 + v0 + lane, my_q);
    }
    const float *qrow = (PAIRS && queries) ? queries + (uint64_t)my_q * q_dim : nullptr;
    float acc[QT];
#pragma unroll
    for (int q = 0; q < QT; ++q) acc[q] = 0.0f;
    for (uint32_t c0 = 0; c0 < dim; c0 += TL1_CH) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int vec = 8 * t + (lane >> 3), e = 2 * (lane & 7);
            double a = 0.0, b = 0.0;
            if (v0 + vec < n) {
                const double *src = deq + (v0 + vec) * padded + c0 + e;
                if (c0 + e < padded) a = src[0];
                if (c0 + e + 1 < padded) b = src[1];
            }
            tile[vec][e] = a;
            tile[vec][e + 1] = b;
        }
        __syncthreads();
        const uint32_t cnt = dim - c0 < TL1_CH ? dim - c0 : TL1_CH;
        for (uint32_t e = 0; e < cnt; ++e) {
            const double v = tile[lane][e];
            if (PAIRS) {
                const double qv = (qrow && live && c0 + e < q_dim) ? (double)qrow[c0 + e] : 0.0;
                acc[0] = acc[0] + (float)__builtin_fabs(qv - v);
            } else {
#pragma unroll
                for (int q = 0; q < QT; ++q) {
                    const double qv = (queries && (uint32_t)q < nq) ? (double)queries[(uint64_t)(q0 + q) * q_dim + c0 + e] : 0.0;
                    acc[q] = acc[q] + (float)__builtin_fabs(qv - v);
                }
            }
        }
    }
    if (!live_row) return;
    if (PAIRS) {
        if (live) scores[col0 + v0 + lane] = invert ? -acc[0] : acc[0];
        return;
    }
#pragma unroll
    for (int q = 0; q < QT; ++q)
        if ((uint32_t)q < nq) scores[(uint64_t)(q0 + q) * stride + col0 + v0 + lane] = invert ? -acc[q] : acc[q];
}

int32_t launch_tq_l1_scores(hipStream_t st, const double *d_deq, uint64_t n, uint32_t padded_dim, uint32_t dim, const float *d_queries, uint32_t q_dim,
                            uint32_t q0, uint32_t nq, float *d_scores, uint64_t stride, uint64_t col0, int invert, const PairSel *sel, uint64_t item0) {
    if (n == 0) return QMX_OK;
    QMX_REQUIRE(dim <= padded_dim && (!d_queries || dim <= q_dim), QMX_ERR_BAD_ARG, "TQ L1: %u coordinates of %u", dim, padded_dim);
    const dim3 grid((uint32_t)((n + 63) / 64)), block(64);
    ::qmx::clear_stale_error();
    if (sel) {
        hipLaunchKernelGGL((tq_l1_score_kernel<1, true>), grid, block, 0, st, d_deq, n, padded_dim, dim, d_queries, q_dim, 0u, 1u, d_scores, stride, col0, invert, *sel, item0);
    } else {
        const PairSel none{nullptr, 0, nullptr};
        for (uint32_t p = 0; p < nq; p += 8) {
            const uint32_t cnt = nq - p < 8 ? nq - p : 8;
            if (cnt == 1) hipLaunchKernelGGL((tq_l1_score_kernel<1, false>), grid, block, 0, st, d_deq, n, padded_dim, dim, d_queries, q_dim, q0 + p, cnt, d_scores, stride, col0, invert, none, item0);
            else if (cnt <= 4) hipLaunchKernelGGL((tq_l1_score_kernel<4, false>), grid, block, 0, st, d_deq, n, padded_dim, dim, d_queries, q_dim, q0 + p, cnt, d_scores, stride, col0, invert, none, item0);
            else hipLaunchKernelGGL((tq_l1_score_kernel<8, false>), grid, block, 0, st, d_deq, n, padded_dim, dim, d_queries, q_dim, q0 + p, cnt, d_scores, stride, col0, invert, none, item0);
        }
    }
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

}  // namespace qmx

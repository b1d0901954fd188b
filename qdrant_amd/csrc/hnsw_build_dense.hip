// hnsw_build_dense.hip — the device HNSW build (hnsw_build.hpp) instantiated for the dense lane policies.
#include "dense_policies.hpp"
#include "hnsw_build.hpp"

namespace qmx {

int32_t launch_hnsw_build_dense(hipStream_t st, int dtype, int distance, const ScanArgs &a, const HnswBuildArgs &h, int phase,
                                uint32_t grid, int *per_cu) {
    const HnswBuildLauncher l{st, &h, phase, grid, per_cu};
    if (dtype == QMX_DTYPE_F32) return dispatch_metric<RowF32, SmallF32, true>(l, distance, a);
    if (dtype == QMX_DTYPE_F16) return dispatch_metric<RowF16, SmallF16, true>(l, distance, a);
    // u8: dot / euclid / manhattan rows are complete query entries; the per-pair cosine (metric_uint/simple_cosine.rs) takes the query's
    // norm from the per-row norm column (RowU8Internal; rows below the AVX threshold compute both norms themselves, SmallU8)
    if (dtype == QMX_DTYPE_U8) {
        QMX_REQUIRE(distance != QMX_DISTANCE_COSINE || a.dim < 32 || (a.row_norms_f && a.row_norms_i), QMX_ERR_BAD_ARG, "u8 cosine build without the row norms");
        return dispatch_metric<RowU8Internal, SmallU8, false>(l, distance, a);
    }
    set_error("device HNSW build: dtype %d with distance %d not supported", dtype, distance);
    return QMX_ERR_NOT_SUPPORTED;
}

// Sum of squares of every stored u8 row, as the cosine leaf computes it for the STORED side of a pair (the query side is computed the
// same way): AVX2 order = 8 exact i32 lanes (lane j = bytes 4j..4j+3 of every 32-byte block), _mm256_cvtepi32_ps, hsum256_ps_avx
// (hi128 + lo128, then (l0 + l1) + (l2 + l3)), + the remainder as one float; scalar order = one exact i32.  One thread per row.
__global__ __launch_bounds__(256) void u8_row_norms_kernel(const unsigned char *rows, uint64_t row_stride, uint64_t n, uint32_t dim, float *norms_f,
                                                           int32_t *norms_i) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const unsigned char *v = rows + r * row_stride;
    const uint32_t body = dim - dim % 32;
    int32_t lane[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rem = 0;
    for (uint32_t b = 0; b < body; b += 32)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t w = *reinterpret_cast<const uint32_t *>(v + b + 4 * j);
            lane[j] = (int32_t)__builtin_amdgcn_udot4(w, w, (uint32_t)lane[j], false);
        }
    for (uint32_t i = body; i < dim; ++i) rem += (int32_t)v[i] * (int32_t)v[i];
    float lr[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) lr[k] = (float)lane[k + 4] + (float)lane[k];          // hi128 + lo128 (dpp row_half_mirror pairs lane k with k + 4)
    float f = (lr[0] + lr[1]) + (lr[2] + lr[3]);
    if (body < dim) f += (float)rem;
    int32_t tot = rem;
#pragma unroll
    for (int j = 0; j < 8; ++j) tot += lane[j];
    norms_f[r] = f;
    norms_i[r] = tot;
}
int32_t launch_u8_row_norms(hipStream_t st, const void *rows, uint64_t row_stride, uint64_t n, uint32_t dim, uint32_t, float *norms_f, int32_t *norms_i) {
    if (n == 0) return QMX_OK;
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(u8_row_norms_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, (const unsigned char *)rows, row_stride, n, dim, norms_f,
                       norms_i);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

}  // namespace qmx

// quantile.hip — the selection step of `find_quantile_interval` (lib/quantization/src/quantile.rs:35-84) on device: the interval
// of the scalar quantizer when `quantile = Some(q)`.
//
// Reference: the values of the sampled vectors are flattened into one slice of `len` floats;
//   cut_index = max(1, min((len - 1) / 2, (n_vectors as f32 * (1.0 - quantile) / 2.0) as usize))          (:58-62; vectors, not values: as is)
//   select_nth_unstable(len - cut_index) keeps sorted positions [0, len - cut_index); select_nth_unstable(cut_index) on that
//   keeps positions (cut_index, len - cut_index); the interval is the min / max of those                   (:63-78)
// = sorted[cut_index + 1] and sorted[len - cut_index - 1].  WHICH vectors are sampled is random in the reference
// (`take_random_vectors`, Permutor): the sample is an input here, as the k-means sample of PQ is.  Given the sample the result
// is a pure order statistic: exact on any implementation (a full radix sort of <= 5 000 x dim floats here).
#include "kernels.hpp"

#include <rocprim/device/device_radix_sort.hpp>

namespace qmx {

// out[0] = sorted[lo_pos], out[1] = sorted[hi_pos] of the n floats at d_in (device); d_tmp: n floats of scratch
int32_t launch_order_statistics_f32(hipStream_t st, const float *d_in, float *d_tmp, uint64_t n, uint64_t lo_pos, uint64_t hi_pos, float *h_out) {
    size_t tmp_bytes = 0;
    ::qmx::clear_stale_error();
    QMX_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, d_in, d_tmp, (size_t)n, 0, 32, st));
    void *d_work = nullptr;
    QMX_HIP(hipMalloc(&d_work, tmp_bytes ? tmp_bytes : 16));
    hipError_t e = rocprim::radix_sort_keys(d_work, tmp_bytes, d_in, d_tmp, (size_t)n, 0, 32, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&h_out[0], d_tmp + lo_pos, sizeof(float), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&h_out[1], d_tmp + hi_pos, sizeof(float), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_work);
    QMX_HIP(e);
    return QMX_OK;
}

}  // namespace qmx

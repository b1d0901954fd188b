// hnsw_build_dense.hip — the device HNSW build (hnsw_build.hpp) instantiated for the dense lane policies.
#include "dense_policies.hpp"
#include "hnsw_build.hpp"

namespace qmx {

int32_t launch_hnsw_build_dense(hipStream_t st, int dtype, int distance, const ScanArgs &a, const HnswBuildArgs &h, int phase,
                                uint32_t grid, int *per_cu) {
    const HnswBuildLauncher l{st, &h, phase, grid, per_cu};
    if (dtype == QMX_DTYPE_F32) return dispatch_metric<RowF32, SmallF32, true>(l, distance, a);
    if (dtype == QMX_DTYPE_F16) return dispatch_metric<RowF16, SmallF16, true>(l, distance, a);
    // u8: dot / euclid / manhattan rows are complete query entries; per-pair cosine (metric_uint/simple_cosine.rs) needs the query's
    // norm in the aux block, which a stored row does not carry
    if (dtype == QMX_DTYPE_U8 && distance != QMX_DISTANCE_COSINE) return dispatch_metric<RowU8, SmallU8, false>(l, distance, a);
    set_error("device HNSW build: dtype %d with distance %d not supported", dtype, distance);
    return QMX_ERR_NOT_SUPPORTED;
}

}  // namespace qmx

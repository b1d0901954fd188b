// hnsw_build_dense.hip — the device HNSW build (hnsw_build.hpp) instantiated for the dense lane policies.
#include "dense_policies.hpp"
#include "hnsw_build.hpp"

namespace qmx {

int32_t launch_hnsw_build_dense(hipStream_t st, int dtype, int distance, const ScanArgs &a, const HnswBuildArgs &h, int phase,
                                uint32_t grid, int *per_cu) {
    // f32 / f16 only: a u8 row cannot stand in for a query entry (per-pair cosine needs the query's norm in the aux block)
    const HnswBuildLauncher l{st, &h, phase, grid, per_cu};
    if (dtype == QMX_DTYPE_F32) return dispatch_metric<RowF32, SmallF32, true>(l, distance, a);
    if (dtype == QMX_DTYPE_F16) return dispatch_metric<RowF16, SmallF16, true>(l, distance, a);
    set_error("device HNSW build: dtype %d not supported", dtype);
    return QMX_ERR_NOT_SUPPORTED;
}

}  // namespace qmx

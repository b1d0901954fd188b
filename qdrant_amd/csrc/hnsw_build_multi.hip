// hnsw_build_multi.hip — the device HNSW build (hnsw_build.hpp) over multi-vector points with dense f32 / f16 inner rows:
// every stored <-> stored score is MaxSim (HopMaxSimInternal, hnsw.hpp).
#include "dense_policies.hpp"
#include "hnsw_build.hpp"

namespace qmx {

int32_t launch_hnsw_build_maxsim_dense(hipStream_t st, int dtype, int distance, const ScanArgs &a, const HnswBuildArgs &h, int phase,
                                       uint32_t grid, int *per_cu) {
    const HnswBuildMaxSimLauncher l{st, &h, phase, grid, per_cu};
    if (dtype == QMX_DTYPE_F32) return dispatch_metric<RowF32, SmallF32, true>(l, distance, a);
    if (dtype == QMX_DTYPE_F16) return dispatch_metric<RowF16, SmallF16, true>(l, distance, a);
    set_error("device HNSW build over multi-vectors: dtype %d with distance %d not supported", dtype, distance);
    return QMX_ERR_NOT_SUPPORTED;
}

}  // namespace qmx

// hnsw_dense.hip — the device-resident HNSW walk (hnsw.hpp) instantiated for the dense f32 / f16 / u8 lane policies.
#include "dense_policies.hpp"
#include "hnsw.hpp"

namespace qmx {

int32_t launch_hnsw_dense(hipStream_t st, int dtype, int distance, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu) {
    return dispatch_dense(HnswLauncher{st, &h, grid, per_cu}, dtype, distance, a);
}
int32_t launch_hnsw_maxsim_dense(hipStream_t st, int dtype, int distance, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu) {
    return dispatch_dense(HnswMaxSimLauncher{st, &h, grid, per_cu}, dtype, distance, a);
}
}  // namespace qmx

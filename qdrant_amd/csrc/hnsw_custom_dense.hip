// hnsw_custom_dense.hip — the device-resident HNSW walk (hnsw.hpp) with a custom query (Recommend / Discover / Context / Feedback) as its scorer,
// instantiated for the dense f32 / f16 / u8 lane policies (CustomQueryScorer, query_scorer/custom_query_scorer.rs:44-121).
#include "dense_policies.hpp"
#include "hnsw.hpp"

namespace qmx {

int32_t launch_hnsw_custom_dense(hipStream_t st, int dtype, int distance, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu) {
    return dispatch_dense(HnswCustomLauncher{st, &h, grid, per_cu}, dtype, distance, a);
}

int32_t launch_hnsw_custom_maxsim_dense(hipStream_t st, int dtype, int distance, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu) {
    return dispatch_dense(HnswCustomMaxSimLauncher{st, &h, grid, per_cu}, dtype, distance, a);
}

}  // namespace qmx

// hnsw_custom_tq.hip - the HNSW walk with a custom query over TurboQuant storage (TurboCustomQueryScorer, query_scorer/turbo_custom_query_scorer.rs:17-113):
// every example is rotated and encoded like a plain query, the walk scores a candidate against each and combines (hnsw.hpp HopCustom).
#include "tq_policies.hpp"

namespace qmx {

int32_t launch_hnsw_custom_tq(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu) {
    return dispatch_tq(HnswCustomLauncher{st, &h, grid, per_cu}, a);
}

}  // namespace qmx

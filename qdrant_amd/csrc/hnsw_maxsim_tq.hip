// hnsw_maxsim_tq.hip - the MaxSim walk over multi-vector points whose inner rows are TurboQuant codes (QuantizedMultivectorStorage<EncodedVectorsTQ>;
// hnsw.hpp HopMaxSim over the RowTQ* policies): the inner query vectors are rotated and encoded like plain queries.
#include "tq_policies.hpp"

namespace qmx {

int32_t launch_hnsw_maxsim_tq(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu) {
    return dispatch_tq(HnswMaxSimLauncher{st, &h, grid, per_cu}, a);
}

}  // namespace qmx

// hnsw_tq.hip - the HNSW walk with the TurboQuant scorer (hnsw.hpp over the RowTQ* policies of tq_policies.hpp).
#include "tq_policies.hpp"

namespace qmx {

int32_t launch_hnsw_tq(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu) {
    return dispatch_tq(HnswLauncher{st, &h, grid, per_cu}, a);
}

}  // namespace qmx

// hnsw_tq1.hip - the HNSW walk with the TurboQuant scorer over 1-bit (and 1.5-bit) storages: 8-bit query values, or the 16-bit ones of TQ+
// (hnsw_tq.hip has the 4- and 2-bit policies and the dispatch).
#include "tq_policies.hpp"

namespace qmx {

int32_t launch_hnsw_tq1(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu) {
    const HnswLauncher l{st, &h, grid, per_cu};
    const bool l2 = a.tq_l2 != nullptr;
    if (a.tq_planes == 16) return l2 ? l.template row<RowTQ1<true, 16>>(a) : l.template row<RowTQ1<false, 16>>(a);
    return l2 ? l.template row<RowTQ1<true>>(a) : l.template row<RowTQ1<false>>(a);
}

}  // namespace qmx

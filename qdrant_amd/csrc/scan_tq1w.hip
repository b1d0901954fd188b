// scan_tq1w.hip - the scan and pair kernels of 1-bit TurboQuant storages under TQ+ (16 query bit planes); see scan_tq1.hip.
#include "tq_policies.hpp"

namespace qmx {

int32_t launch_scan_tq1_wide(hipStream_t st, int qt, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    const ScanLauncher l{st, qt, mode, num_cus, grid_out};
    return a.tq_l2 ? l.template row<RowTQ1<true, 16>>(a) : l.template row<RowTQ1<false, 16>>(a);
}
int32_t launch_pairs_tq1_wide(hipStream_t st, const ScanArgs &a, const PairSel &sel, uint64_t n_items, int num_cus) {
    const PairLauncher l{st, sel, n_items, num_cus};
    return a.tq_l2 ? l.template row<RowTQ1<true, 16>>(a) : l.template row<RowTQ1<false, 16>>(a);
}

}  // namespace qmx

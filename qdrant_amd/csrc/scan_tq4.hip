// scan_tq4.hip - the brute-force scan and pair kernels of 4-bit TurboQuant storages (policies: tq_policies.hpp; dispatch: scan_tq.hip).
#include "tq_policies.hpp"

namespace qmx {

int32_t launch_scan_tq4(hipStream_t st, int qt, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    const ScanLauncher l{st, qt, mode, num_cus, grid_out};
    return a.tq_l2 ? l.template row<RowTQ4<true>>(a) : l.template row<RowTQ4<false>>(a);
}
int32_t launch_pairs_tq4(hipStream_t st, const ScanArgs &a, const PairSel &sel, uint64_t n_items, int num_cus) {
    const PairLauncher l{st, sel, n_items, num_cus};
    return a.tq_l2 ? l.template row<RowTQ4<true>>(a) : l.template row<RowTQ4<false>>(a);
}

}  // namespace qmx

// scan_tq1.hip - the brute-force scan and pair kernels of 1-bit TurboQuant storages (policies: tq_policies.hpp; dispatch: scan_tq.hip).
#include "tq_policies.hpp"

namespace qmx {

// 16-bit query planes (TQ+ over 1-bit storage): scan_tq1w.hip
int32_t launch_scan_tq1_wide(hipStream_t st, int qt, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid_out);
int32_t launch_pairs_tq1_wide(hipStream_t st, const ScanArgs &a, const PairSel &sel, uint64_t n_items, int num_cus);

int32_t launch_scan_tq1(hipStream_t st, int qt, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    const ScanLauncher l{st, qt, mode, num_cus, grid_out};
    if (a.tq_planes == 16) return launch_scan_tq1_wide(st, qt, mode, a, num_cus, grid_out);
    return a.tq_l2 ? l.template row<RowTQ1<true>>(a) : l.template row<RowTQ1<false>>(a);
}
int32_t launch_pairs_tq1(hipStream_t st, const ScanArgs &a, const PairSel &sel, uint64_t n_items, int num_cus) {
    const PairLauncher l{st, sel, n_items, num_cus};
    if (a.tq_planes == 16) return launch_pairs_tq1_wide(st, a, sel, n_items, num_cus);
    return a.tq_l2 ? l.template row<RowTQ1<true>>(a) : l.template row<RowTQ1<false>>(a);
}

}  // namespace qmx

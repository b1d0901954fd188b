// scan_dense.hip — brute-force scan and pair-scoring launchers of the dense f32 / f16 / u8 lane policies
// (dense_policies.hpp); the HNSW walk and build over the same policies live in hnsw_dense.hip / hnsw_build_dense.hip
// (separate translation units: they compile in parallel).
#include "dense_policies.hpp"

namespace qmx {

int32_t launch_scan_dense(hipStream_t st, int dtype, int distance, int qt, ScanMode mode,
                          const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    return dispatch_dense(ScanLauncher{st, qt, mode, num_cus, grid_out}, dtype, distance, a);
}
int32_t launch_pairs_dense(hipStream_t st, int dtype, int distance, const ScanArgs &a, const PairSel &sel,
                           uint64_t n_items, int num_cus) {
    return dispatch_dense(PairLauncher{st, sel, n_items, num_cus}, dtype, distance, a);
}
}  // namespace qmx

// hnsw_tq_l1.hip - the HNSW walk through a TurboQuant storage over Manhattan (tq_l1_policy.hpp): rotations of a multiple of 16 / 32 / 64 coordinates.
#include "tq_l1_policy.hpp"

namespace qmx {

int32_t launch_hnsw_tq_l1(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu, uint32_t rot_dim) {
    if (rot_dim % 16 == 0 && rot_dim <= 1024) return launch_hnsw_hop<HopTQL1<16>>(st, a, h, grid, per_cu);
    if (rot_dim % 32 == 0 && rot_dim <= 2048) return launch_hnsw_hop<HopTQL1<32>>(st, a, h, grid, per_cu);
    if (rot_dim % 64 == 0 && rot_dim <= 4096) return launch_hnsw_hop<HopTQL1<64>>(st, a, h, grid, per_cu);
    set_error("HNSW walk through a TurboQuant storage over Manhattan: a rotation over %u coordinates is not a multiple of 16 (up to 1024), 32 (2048) or 64 (4096)", rot_dim);
    return QMX_ERR_NOT_SUPPORTED;
}

// ... with a custom query (TurboCustomQueryScorer over such a storage): the hop against every example, combined per candidate (hnsw.hpp HopCustom::hop)
int32_t launch_hnsw_custom_tq_l1(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu, uint32_t rot_dim) {
    if (rot_dim % 16 == 0 && rot_dim <= 1024) return launch_hnsw_hop<HopCustom<HopTQL1<16>>>(st, a, h, grid, per_cu);
    if (rot_dim % 32 == 0 && rot_dim <= 2048) return launch_hnsw_hop<HopCustom<HopTQL1<32>>>(st, a, h, grid, per_cu);
    if (rot_dim % 64 == 0 && rot_dim <= 4096) return launch_hnsw_hop<HopCustom<HopTQL1<64>>>(st, a, h, grid, per_cu);
    set_error("custom HNSW walk through a TurboQuant storage over Manhattan: a rotation over %u coordinates is not a multiple of 16 (up to 1024), 32 (2048) or 64 (4096)",
              rot_dim);
    return QMX_ERR_NOT_SUPPORTED;
}

}  // namespace qmx

// hnsw_tq.hip - the HNSW walk with the TurboQuant scorer (hnsw.hpp over the RowTQ* policies of tq_policies.hpp): 4- and 2-bit storages here,
// 1-bit storages in hnsw_tq1.hip (two translation units: each policy is five kernels - register beams of 128 / 512 entries with the query entry in
// LDS or read through L2, and the LDS beam of wider searches - and the four policies of one file are what a compiler job should carry).
#include "tq_policies.hpp"

namespace qmx {

int32_t launch_hnsw_tq1(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu);   // hnsw_tq1.hip

int32_t launch_hnsw_tq(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu) {
    const HnswLauncher l{st, &h, grid, per_cu};
    const bool l2 = a.tq_l2 != nullptr;
    if (a.tq_bits == 4) return l2 ? l.template row<RowTQ4<true>>(a) : l.template row<RowTQ4<false>>(a);
    if (a.tq_bits == 2) return l2 ? l.template row<RowTQ2<true>>(a) : l.template row<RowTQ2<false>>(a);
    if (a.tq_bits == 1) return launch_hnsw_tq1(st, a, h, grid, per_cu);
    set_error("TurboQuant: %u bits per value not supported", a.tq_bits);
    return QMX_ERR_NOT_SUPPORTED;
}

}  // namespace qmx

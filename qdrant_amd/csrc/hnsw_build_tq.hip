// hnsw_build_tq.hip - the HNSW build through the TurboQuant scorer (hnsw_build.hpp).
#include "tq_internal_policy.hpp"

namespace qmx {

// ------------------------------------------------------------------------------------------
// HNSW build over a TurboQuant segment (hnsw/build.rs:334-341 + point_scorer.rs:183-218).  EncodedVectorsTQ cannot turn a stored row into a
// query (encode_internal_vector -> None), so - as for PQ - the searches of an insertion score through precompute_query of the point's ORIGINAL
// vector (the RowTQ* policies over the batch's entries, made by api_hnsw.hip before phase 1 and staged in LDS per insertion) while everything
// stored <-> stored - the heuristic, the back links, an entry point at or below the new point's level - is score_symmetric
// (turboquant/quantization.rs:395-494): the integer dot of the two rows' codebook bytes (1 bit: dim - 2 popcount(a ^ b)), as tq_internal_kernel
// below.  One lane per stored row; `qp` is the other row's code bytes inside the block, so its index - the extras columns - follows from the pointer.
// ------------------------------------------------------------------------------------------
int32_t launch_hnsw_build_tq(hipStream_t st, const ScanArgs &a, const HnswBuildArgs &h, int phase, uint32_t grid, int *per_cu) {
    QMX_REQUIRE(h.batch_queries, QMX_ERR_BAD_ARG, "TurboQuant build needs the batch's query entries");
    const bool l2 = a.tq_l2 != nullptr;
#define QMX_TQB(B, L, ROW)                                                                                                      \
    if (a.tq_bits == B && l2 == L) return launch_hnsw_build_hop<HopRow<ROW>, HopTQInternal<B, L>>(st, a, h, phase, grid, per_cu);
    QMX_TQB(4, false, RowTQ4<false>)
    QMX_TQB(4, true, RowTQ4<true>)
    QMX_TQB(2, false, RowTQ2<false>)
    QMX_TQB(2, true, RowTQ2<true>)
#undef QMX_TQB
    if (a.tq_bits == 1) {
        if (a.tq_planes == 16)
            return l2 ? launch_hnsw_build_hop<HopRow<RowTQ1<true, 16>>, HopTQInternal<1, true>>(st, a, h, phase, grid, per_cu)
                      : launch_hnsw_build_hop<HopRow<RowTQ1<false, 16>>, HopTQInternal<1, false>>(st, a, h, phase, grid, per_cu);
        return l2 ? launch_hnsw_build_hop<HopRow<RowTQ1<true>>, HopTQInternal<1, true>>(st, a, h, phase, grid, per_cu)
                  : launch_hnsw_build_hop<HopRow<RowTQ1<false>>, HopTQInternal<1, false>>(st, a, h, phase, grid, per_cu);
    }
    set_error("TurboQuant build: %u bits per value not supported", a.tq_bits);
    return QMX_ERR_NOT_SUPPORTED;
}

}  // namespace qmx

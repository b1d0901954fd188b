// hnsw_build_multi_tq.hip - the HNSW build over multi-vector points with TurboQuant inner rows: no stored row is a query (EncodedVectorsTQ::
// encode_internal_vector -> None, so QuantizedMultivectorStorage's is None too, quantized_multivector_storage/mod.rs:458-470): the searches of an
// insertion score through the precomputed queries of the point's ORIGINAL inner vectors (HopMaxSimQ over the batch's entries), stored <-> stored pairs
// through score_internal_max_similarity over score_symmetric (HopMaxSimInternal over HopTQInternal).
#include "hnsw_build.hpp"
#include "tq_internal_policy.hpp"

namespace qmx {

int32_t launch_hnsw_build_maxsim_tq(hipStream_t st, const ScanArgs &a, const HnswBuildArgs &h, int phase, uint32_t grid, int *per_cu) {
    QMX_REQUIRE(h.batch_queries && a.mv_offsets, QMX_ERR_BAD_ARG, "multi-vector TurboQuant build needs the batch's query entries and the point offsets");
    const bool l2 = a.tq_l2 != nullptr;
#define QMX_TQMB(B, L, ROW)                                                                                                      \
    if (a.tq_bits == B && l2 == L) return launch_hnsw_build_hop<HopMaxSimQ<HopRow<ROW>>, HopMaxSimInternal<HopTQInternal<B, L>>>(st, a, h, phase, grid, per_cu);
    QMX_TQMB(4, false, RowTQ4<false>)
    QMX_TQMB(4, true, RowTQ4<true>)
    QMX_TQMB(2, false, RowTQ2<false>)
    QMX_TQMB(2, true, RowTQ2<true>)
#undef QMX_TQMB
    if (a.tq_bits == 1) {
        if (a.tq_planes == 16)
            return l2 ? launch_hnsw_build_hop<HopMaxSimQ<HopRow<RowTQ1<true, 16>>>, HopMaxSimInternal<HopTQInternal<1, true>>>(st, a, h, phase, grid, per_cu)
                      : launch_hnsw_build_hop<HopMaxSimQ<HopRow<RowTQ1<false, 16>>>, HopMaxSimInternal<HopTQInternal<1, false>>>(st, a, h, phase, grid, per_cu);
        return l2 ? launch_hnsw_build_hop<HopMaxSimQ<HopRow<RowTQ1<true>>>, HopMaxSimInternal<HopTQInternal<1, true>>>(st, a, h, phase, grid, per_cu)
                  : launch_hnsw_build_hop<HopMaxSimQ<HopRow<RowTQ1<false>>>, HopMaxSimInternal<HopTQInternal<1, false>>>(st, a, h, phase, grid, per_cu);
    }
    set_error("TurboQuant build: %u bits per value not supported", a.tq_bits);
    return QMX_ERR_NOT_SUPPORTED;
}

}  // namespace qmx

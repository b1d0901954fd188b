// hnsw_build_tq_l1.hip - the HNSW build through a TurboQuant storage over Manhattan (hnsw_build.hpp + tq_l1_policy.hpp).
#include "hnsw_build.hpp"
#include "tq_l1_policy.hpp"

namespace qmx {

// As every TurboQuant build (hnsw_build_tq.hip): EncodedVectorsTQ cannot turn a stored row into a query, so the searches of an insertion score through the
// query scorer of the point's ORIGINAL vector - over Manhattan that is the vector as given against the dequantised, back-rotated candidate
// (score_precomputed's L1 arm, quantization.rs:596-607: HopTQL1) - while stored <-> stored pairs are score_symmetric's L1 arm (:429-440: HopTQL1Internal).
// The entries of a batch are its original rows, tq_l1_query_bytes(dim) apart; the hop scratch of HopTQL1 lies behind the staged entry.
int32_t launch_hnsw_build_tq_l1(hipStream_t st, const ScanArgs &a, const HnswBuildArgs &h, int phase, uint32_t grid, int *per_cu, uint32_t rot_dim,
                                uint32_t padded_dim) {
    QMX_REQUIRE(h.batch_queries, QMX_ERR_BAD_ARG, "TurboQuant build needs the batch's query entries");
    const uint32_t hi = rot_dim > padded_dim ? rot_dim : padded_dim;
    {   // static LDS of HopTQL1Internal<E> (rotation buffer + parked terms) + the launch's dynamic share (the entry and HopTQL1's scratch behind it, lists)
        const uint32_t e = hi <= 1024 ? 16 : hi <= 2048 ? 32 : 64, g = e == 16 ? 8 : e == 32 ? 4 : 1;
        const size_t need = (size_t)64 * e * 8 + (size_t)g * (64 * e + 1) * 4 + h.lds_query_bytes + 8 * 1024 + 8 * (size_t)h.ef_construct;
        QMX_REQUIRE(need <= 160 * 1024, QMX_ERR_NOT_SUPPORTED, "HNSW build through TurboQuant over Manhattan: %u coordinates need %zu bytes of LDS per insertion", hi, need);
    }
    if (rot_dim % 16 == 0 && hi <= 1024) return launch_hnsw_build_hop<HopTQL1<16>, HopTQL1Internal<16>>(st, a, h, phase, grid, per_cu);
    if (rot_dim % 32 == 0 && hi <= 2048) return launch_hnsw_build_hop<HopTQL1<32>, HopTQL1Internal<32>>(st, a, h, phase, grid, per_cu);
    if (rot_dim % 64 == 0 && hi <= 4096) return launch_hnsw_build_hop<HopTQL1<64>, HopTQL1Internal<64>>(st, a, h, phase, grid, per_cu);
    set_error("HNSW build through a TurboQuant storage over Manhattan: a rotation over %u of %u coordinates is not a multiple of 16 (up to 1024), 32 (2048) or 64 (4096)",
              rot_dim, padded_dim);
    return QMX_ERR_NOT_SUPPORTED;
}

}  // namespace qmx

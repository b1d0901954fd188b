// scan_bq.hip — EncodedVectorsBin (binary quantization) on device: Encoding::OneBit, QueryEncoding::SameAsStorage,
// BitsStoreType = u128 (what single-vector segments use, vector_storage/quantized/quantized_vectors/binary/create.rs:34-37).
//
// Reference (lib/quantization/src/encoded_vectors_binary.rs):
//   encode_one_bit_vector   :558-568   bit i = vector[i] > 0.0, little-endian inside u128 words
//   storage size            :829-840 + BitsStoreType::get_storage_size for u128 :412-419  (ceil(dim / 128) * 16 bytes)
//   xor_popcnt (u128)       :288-333 -> cpp/sse.c:54-75 impl_xor_popcnt_sse_uint128
//   calculate_metric        :766-810   xor = popcount(v ^ q); zeros = dim - xor; (Dot | Cosine, !invert) and (L1 | L2, invert)
//                                      -> zeros - xor  (invert derives from the distance, quantized_vectors.rs:232)
//   score_internal          :892-917   the same metric between two stored rows; encode_internal_vector :923-934 = the row itself
// Integer work: the popcount is exact in any order, dim and xor are < 2^24, so the f32 arithmetic of calculate_metric is
// exact: scores are bit-identical to the reference's.
//
// The rows are 16-byte multiples: the lane policies of the dense scan apply unchanged (8 lanes per row, 16 bytes per
// lane per 128-byte step) and with them the tiled scan, pair scoring, rescoring plumbing and the HNSW walk.
#include "hnsw.hpp"

namespace qmx {

struct RowBQ {
    static constexpr int NACC = 1;
    static constexpr int NRAUX = 0;
    static constexpr int R16 = 2;
    typedef uint32_t acc_t;
    static __device__ __forceinline__ void row_aux(acc_t (&)[1], const uint4 &) {}
    static __device__ __forceinline__ void mac(acc_t (&a)[NACC], const uint4 &q, const uint4 &v) {
        a[0] += (uint32_t)(__popc(q.x ^ v.x) + __popc(q.y ^ v.y) + __popc(q.z ^ v.z) + __popc(q.w ^ v.w));
    }
    static __device__ __forceinline__ float finish(acc_t (&a)[NACC], acc_t (&)[1], const unsigned char *, const unsigned char *, uint32_t,
                                                   const ScanArgs &args) {
        const float xor_product = (float)reduce8_u32(a[0]);
        const float dim = (float)args.bq_dim;
        const float zeros_count = dim - xor_product;
        return args.bq_flip ? xor_product - zeros_count : zeros_count - xor_product;
    }
};

template <class L>
static int32_t dispatch_bq(const L &l, const ScanArgs &a) { return l.template row<RowBQ>(a); }

int32_t launch_scan_bq(hipStream_t st, int qt, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    return dispatch_bq(ScanLauncher{st, qt, mode, num_cus, grid_out}, a);
}
int32_t launch_pairs_bq(hipStream_t st, const ScanArgs &a, const PairSel &sel, uint64_t n_items, int num_cus) {
    return dispatch_bq(PairLauncher{st, sel, n_items, num_cus}, a);
}
int32_t launch_hnsw_bq(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu) {
    return dispatch_bq(HnswLauncher{st, &h, grid, per_cu}, a);
}

// encode_one_bit_vector for a batch: in [n][dim] f32 -> out [n][out_stride] bytes (row_bytes = ceil(dim / 128) * 16, rest of
// the stride zero).  One thread per output dword.
__global__ __launch_bounds__(256) void bq_encode_kernel(const float *in, uint64_t n, uint32_t dim, uint32_t row_bytes, uint8_t *out,
                                                        uint64_t out_stride) {
    const uint32_t words = row_bytes / 4;
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t r = gid / words;
    const uint32_t w = (uint32_t)(gid % words);
    if (r >= n) return;
    uint32_t bits = 0;
    for (uint32_t b = 0; b < 32; ++b) {
        const uint32_t i = w * 32 + b;
        if (i < dim && in[r * dim + i] > 0.0f) bits |= 1u << b;
    }
    *reinterpret_cast<uint32_t *>(out + r * out_stride + (uint64_t)w * 4) = bits;
}
int32_t launch_bq_encode(hipStream_t st, const float *d_in, uint64_t n, uint32_t dim, uint8_t *d_out, uint64_t out_stride) {
    if (n == 0) return QMX_OK;
    const uint32_t row_bytes = ((dim + 127) / 128) * 16;
    const uint64_t total = n * (row_bytes / 4);
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(bq_encode_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, d_in, n, dim, row_bytes, d_out, out_stride);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

}  // namespace qmx

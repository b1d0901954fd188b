// scan_quant.hip — EncodedVectorsU8 (scalar int8 quantization) on device.
//
// Reference (lib/quantization):
//   score            src/encoded_vectors_u8.rs:471-490 (score_point_avx) -> cpp/avx2.c:25-63 (impl_score_dot_avx),
//                    cpp/avx2.c:65-122 (impl_score_l1_avx), postprocess_score :100-103
//   encode           src/encoded_vectors_u8.rs:94-98 (encode_value), :236-296 (row loop), :116-134 (get_shift)
//   encode_query     src/encoded_vectors_u8.rs:583-619
//   score_internal   src/encoded_vectors_u8.rs:675-705, postprocess_internal_score :105-114
// Integer work: everything here is bit-exact against the x86 AVX2 path.
//
// HBM layout: the reference row `[f32 vector_offset][u8 code x actual_dim]` (772 B at d=768, only
// 4-byte aligned) is split at upload into a 16-byte aligned code block [n][actual_dim] and an
// offset column [n] f32; algorithmic bytes per scored row stay 4 + actual_dim.
#include "hnsw_build.hpp"

namespace qmx {

// codes are <= 127, so u8 x u8 dot4 is exact.  AVX2 lane map (avx2.c:41-45): maddubs pairs bytes
// (2m, 2m+1) into i16 lane m; cvtepi16_epi32 of the low / high half both land on i32 lane m % 8,
// i.e. byte pair j' of EITHER 16-byte half goes to lane j' — the same for every 16-byte piece
// (and for the 16-byte tail block, :49-59).
//   SEP = false: 127^2 * actual_dim < 2^24 — every lane sum and every f32 add of the reference is
//                exact, the score equals (float)(total) whatever the order: one dot4 per dword.
//   SEP = true : lane sums stay exact but the f32 hsum may round: keep the 8 lanes apart
//                (two masked dot4 per dword) and add them in HSUM256_PS order (avx2.c:7-14).
template <bool L1, bool SEP>
struct RowSQ {
    static constexpr int NACC = L1 ? 1 : (SEP ? 8 : 4);
    static constexpr int NRAUX = 0;
    static constexpr int R16 = SEP ? 1 : 2;
    typedef uint32_t acc_t;

    static __device__ __forceinline__ void row_aux(acc_t (&)[1], const uint4 &) {}
    static __device__ __forceinline__ void mac(acc_t (&a)[NACC], const uint4 &q, const uint4 &v) {
        const uint32_t qq[4] = {q.x, q.y, q.z, q.w}, vv[4] = {v.x, v.y, v.z, v.w};
        if (L1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) a[0] = __builtin_amdgcn_sad_u8(qq[k], vv[k], a[0]);
        } else if (!SEP) {
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = __builtin_amdgcn_udot4(qq[k], vv[k], a[k], false);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a[2 * k] = __builtin_amdgcn_udot4(qq[k] & 0x0000FFFFu, vv[k], a[2 * k], false);
                a[2 * k + 1] = __builtin_amdgcn_udot4(qq[k] & 0xFFFF0000u, vv[k], a[2 * k + 1], false);
            }
        }
    }
    static __device__ __forceinline__ float finish(acc_t (&a)[NACC], acc_t (&)[1], const unsigned char *q_lds,
                                                   const unsigned char *, uint32_t rid, const ScanArgs &args) {
        float f;
        if (L1) {
            f = (float)(int32_t)reduce8_u32(a[0]);                       // HSUM256_EPI32 then (float)sum
        } else if (!SEP) {
            f = (float)(int32_t)reduce8_u32((a[0] + a[1]) + (a[2] + a[3]));
        } else {
            float l[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) l[j] = (float)(int32_t)reduce8_u32(a[j]);   // _mm256_cvtepi32_ps
            const float x0 = l[4] + l[0], x1 = l[5] + l[1], x2 = l[6] + l[2], x3 = l[7] + l[3];   // hi128 + lo128
            f = (x0 + x2) + (x1 + x3);                                               // movehl add, then add_ss
        }
        const QueryAux *aux = reinterpret_cast<const QueryAux *>(q_lds + args.aux_off);
        // postprocess_score: multiplier * score + query_offset + vector_offset, left to right, not fused
        const float m = args.sq_multiplier * f;
        const float mq = m + aux->f0;
        return mq + args.row_offsets[rid];
    }
};

// RowSQ for the walk that brings the rows' offsets itself (hnsw.hpp: the level-0 link rows of an SQ graph carry the linked rows' vector_offset next to
// their ids, so a hop's offsets arrive with its links - one more line of the link row instead of one random 4-byte request per scored row; the offsets
// column cost the walk a fifth of its time, profiles/r6_sq_walk_offsets.md).  finish stops before the last add: multiplier * score + query_offset; the
// caller adds vector_offset - the same three operations in the same order (postprocess_score, encoded_vectors_u8.rs:100-103).
template <bool L1, bool SEP>
struct RowSQX : RowSQ<L1, SEP> {
    typedef RowSQ<L1, SEP> B;
    static constexpr bool OFFSET_BY_CALLER = true;
    static __device__ __forceinline__ float finish(typename B::acc_t (&a)[B::NACC], typename B::acc_t (&)[1], const unsigned char *q_lds,
                                                   const unsigned char *, uint32_t, const ScanArgs &args) {
        float f;
        if (L1) {
            f = (float)(int32_t)reduce8_u32(a[0]);
        } else if (!SEP) {
            f = (float)(int32_t)reduce8_u32((a[0] + a[1]) + (a[2] + a[3]));
        } else {
            float l[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) l[j] = (float)(int32_t)reduce8_u32(a[j]);
            const float x0 = l[4] + l[0], x1 = l[5] + l[1], x2 = l[6] + l[2], x3 = l[7] + l[3];
            f = (x0 + x2) + (x1 + x3);
        }
        const QueryAux *aux = reinterpret_cast<const QueryAux *>(q_lds + args.aux_off);
        const float m = args.sq_multiplier * f;
        return m + aux->f0;
    }
};

// The same row policy with a STORED ROW as the query (HNSW build): the query entry is a bare code row, its offset
// (vector_offset - shift, postprocess_internal_score :105-114) arrives in ScanArgs::sq_qoff (hnsw_build.hpp query_args).
template <bool L1, bool SEP>
struct RowSQInternal : RowSQ<L1, SEP> {
    typedef RowSQ<L1, SEP> B;
    static constexpr bool INTERNAL_QOFF = true;
    static __device__ __forceinline__ float finish(typename B::acc_t (&a)[B::NACC], typename B::acc_t (&)[1], const unsigned char *,
                                                   const unsigned char *, uint32_t rid, const ScanArgs &args) {
        float f;
        if (L1) {
            f = (float)(int32_t)reduce8_u32(a[0]);
        } else if (!SEP) {
            f = (float)(int32_t)reduce8_u32((a[0] + a[1]) + (a[2] + a[3]));
        } else {
            float l[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) l[j] = (float)(int32_t)reduce8_u32(a[j]);
            const float x0 = l[4] + l[0], x1 = l[5] + l[1], x2 = l[6] + l[2], x3 = l[7] + l[3];
            f = (x0 + x2) + (x1 + x3);
        }
        const float m = args.sq_multiplier * f;
        const float mq = m + args.sq_qoff;
        return mq + args.row_offsets[rid];
    }
};

template <class L>
static int32_t dispatch_sq(const L &l, int distance, const ScanArgs &a) {
    const bool l1 = distance == QMX_DISTANCE_MANHATTAN;
    const bool sep = (uint64_t)127 * 127 * a.dim >= (1ull << 24);
    if (l1) return l.template row<RowSQ<true, false>>(a);
    if (sep) return l.template row<RowSQ<false, true>>(a);
    return l.template row<RowSQ<false, false>>(a);
}
int32_t launch_scan_sq(hipStream_t st, int distance, int qt, ScanMode mode, const ScanArgs &a, int num_cus,
                       uint32_t *grid_out) {
    return dispatch_sq(ScanLauncher{st, qt, mode, num_cus, grid_out}, distance, a);
}
int32_t launch_pairs_sq(hipStream_t st, int distance, const ScanArgs &a, const PairSel &sel, uint64_t n_items, int num_cus) {
    return dispatch_sq(PairLauncher{st, sel, n_items, num_cus}, distance, a);
}
int32_t launch_hnsw_sq(hipStream_t st, int distance, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu) {
    if (h.l0_aux_off) {      // the link rows carry the offsets (api_hnsw.hip: plain walks over a packed level 0)
        const HnswLauncher l{st, &h, grid, per_cu};
        if (distance == QMX_DISTANCE_MANHATTAN) return l.template row<RowSQX<true, false>>(a);
        if ((uint64_t)127 * 127 * a.dim >= (1ull << 24)) return l.template row<RowSQX<false, true>>(a);
        return l.template row<RowSQX<false, false>>(a);
    }
    return dispatch_sq(HnswLauncher{st, &h, grid, per_cu}, distance, a);
}
int32_t launch_hnsw_custom_sq(hipStream_t st, int distance, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu) {
    return dispatch_sq(HnswCustomLauncher{st, &h, grid, per_cu}, distance, a);
}
int32_t launch_hnsw_custom_maxsim_sq(hipStream_t st, int distance, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu) {
    return dispatch_sq(HnswCustomMaxSimLauncher{st, &h, grid, per_cu}, distance, a);
}
int32_t launch_hnsw_maxsim_sq(hipStream_t st, int distance, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu) {
    return dispatch_sq(HnswMaxSimLauncher{st, &h, grid, per_cu}, distance, a);
}
int32_t launch_hnsw_build_sq(hipStream_t st, int distance, const ScanArgs &a, const HnswBuildArgs &h, int phase, uint32_t grid, int *per_cu) {
    const HnswBuildLauncher l{st, &h, phase, grid, per_cu};
    const bool l1 = distance == QMX_DISTANCE_MANHATTAN;
    const bool sep = (uint64_t)127 * 127 * a.dim >= (1ull << 24);
    if (l1) return l.template row<RowSQInternal<true, false>>(a);
    if (sep) return l.template row<RowSQInternal<false, true>>(a);
    return l.template row<RowSQInternal<false, false>>(a);
}
// ... over multi-vector points whose inner rows are SQ codes (score_internal_max_similarity, quantized_multivector_storage/mod.rs:366-393)
int32_t launch_hnsw_build_maxsim_sq(hipStream_t st, int distance, const ScanArgs &a, const HnswBuildArgs &h, int phase, uint32_t grid, int *per_cu) {
    const HnswBuildMaxSimLauncher l{st, &h, phase, grid, per_cu};
    const bool l1 = distance == QMX_DISTANCE_MANHATTAN;
    const bool sep = (uint64_t)127 * 127 * a.dim >= (1ull << 24);
    if (l1) return l.template row<RowSQInternal<true, false>>(a);
    if (sep) return l.template row<RowSQInternal<false, true>>(a);
    return l.template row<RowSQInternal<false, false>>(a);
}

// ------------------------------------------------------------------------------------------
// encode
// ------------------------------------------------------------------------------------------
struct SqEnc {
    float alpha, offset;
    uint32_t dim, actual_dim;
    int is_dot;   // Dot | Cosine
    int is_l2;
    int invert;
};

__device__ __forceinline__ uint8_t sq_encode_value(const SqEnc &p, float value) {
    float i = (value - p.offset) / p.alpha;          // IEEE divide (encoded_vectors_u8.rs:95)
    if (i != i) return 0;                             // clamp keeps NaN, round keeps NaN, `as u8` -> 0
    i = i < 0.0f ? 0.0f : (i > 127.0f ? 127.0f : i);
    return (uint8_t)__builtin_roundf(i);              // f32::round: half away from zero
}

// sum of the codes (Dot) or of their squares (L2) as the reference's sequential f32 sum.  While the
// total stays below 2^24 every partial sum is an exact integer, so the integer wave reduction of
// `part` (each lane's own codes) gives the same bits; beyond that every lane repeats the sequential
// f32 loop over the codes just written (after an agent-scope fence).
__device__ __forceinline__ float sq_codes_offset(const SqEnc &p, const uint8_t *codes, uint32_t part) {
    float off = 0.0f;
    if (p.is_dot || p.is_l2) {
        const uint64_t bound = p.is_dot ? (uint64_t)127 * p.actual_dim : (uint64_t)127 * 127 * p.actual_dim;
        float s;
        if (bound < (1ull << 24)) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
            s = (float)part;
        } else {
            __threadfence();
            s = -0.0f;
            for (uint32_t i = 0; i < p.actual_dim; ++i) {
                const float c = (float)__builtin_nontemporal_load(codes + i);
                s += p.is_dot ? c : c * c;
            }
        }
        off = p.is_dot ? s * p.alpha * p.offset : s * p.alpha * p.alpha;   // left to right
    }
    return p.invert ? -off : off;
}

__device__ __forceinline__ float sq_shift(const SqEnc &p) {               // get_shift :116-134
    const float shift = p.is_dot ? (float)p.actual_dim * p.offset * p.offset : 0.0f;
    return p.invert ? -shift : shift;
}

// One wavefront per vector.  mode 0: stored row -> codes[n][actual_dim] + offsets[n] (or, when
// `rows_out` is given, the reference row layout [f32][codes]); mode 1: query -> tile entry + aux.
__global__ __launch_bounds__(256) void sq_encode_kernel(SqEnc p, const float *in, uint64_t n, uint8_t *codes_out,
                                                        uint64_t codes_stride, float *offsets_out, uint8_t *rows_out,
                                                        int is_query, uint32_t aux_off) {
    const int lane = threadIdx.x & 63;
    const uint64_t vec = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (vec >= n) return;
    const float *v = in + vec * p.dim;
    uint8_t *codes = rows_out ? rows_out + vec * (4 + (uint64_t)p.actual_dim) + 4 : codes_out + vec * codes_stride;
    const float placeholder = p.is_dot ? 0.0f : p.offset;                 // :246-255, :586-595
    uint32_t part = 0;
    for (uint32_t i = lane; i < p.actual_dim; i += 64) {
        const uint32_t c = sq_encode_value(p, i < p.dim ? v[i] : placeholder);
        codes[i] = (uint8_t)c;
        part += p.is_dot ? c : c * c;
    }
    float off = sq_codes_offset(p, codes, part);
    if (!is_query) off = sq_shift(p) + off;                                // :281-283
    if (lane == 0) {
        if (rows_out) *reinterpret_cast<float *>(rows_out + vec * (4 + (uint64_t)p.actual_dim)) = off;   // 4-byte aligned: row size is 4 + 16k
        else if (is_query) reinterpret_cast<QueryAux *>(codes + aux_off)->f0 = off;
        else offsets_out[vec] = off;
    }
}

int32_t launch_sq_encode(hipStream_t st, int distance, const qmx_sq_params &sp, uint32_t dim, const float *in, uint64_t n,
                         uint8_t *codes_out, uint64_t codes_stride, float *offsets_out, uint8_t *rows_out, int is_query,
                         uint32_t aux_off) {
    if (n == 0) return QMX_OK;
    SqEnc p;
    p.alpha = sp.alpha;
    p.offset = sp.offset;
    p.dim = dim;
    p.actual_dim = sp.actual_dim;
    p.is_dot = distance == QMX_DISTANCE_DOT || distance == QMX_DISTANCE_COSINE;
    p.is_l2 = distance == QMX_DISTANCE_EUCLID;
    p.invert = sp.invert;
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(sq_encode_kernel, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, st, p, in, n, codes_out, codes_stride,
                       offsets_out, rows_out, is_query, aux_off);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

// reference rows [f32 offset][codes] <-> SoA (codes block + offset column); one wavefront per row
__global__ __launch_bounds__(256) void sq_split_kernel(const uint8_t *rows, uint64_t row_stride, uint64_t n, uint32_t actual_dim,
                                                       uint8_t *codes, float *offsets, int to_rows, const uint32_t *ids,
                                                       uint64_t n_rows, int *err_flag) {
    const int lane = threadIdx.x & 63;
    const uint64_t w = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= n) return;
    if (!to_rows) {
        const uint8_t *src = rows + w * row_stride;
        if (lane == 0) {
            float f;
            memcpy(&f, src, 4);
            offsets[w] = f;
        }
        for (uint32_t i = lane; i < actual_dim; i += 64) codes[w * actual_dim + i] = src[4 + i];
    } else {   // gather rows `ids` back into the reference layout (get_quantized_vector)
        const uint32_t id = ids[w];
        if (id >= n_rows) {
            if (lane == 0) *err_flag = 1;
            return;
        }
        uint8_t *dst = const_cast<uint8_t *>(rows) + w * row_stride;
        if (lane == 0) {
            const float f = offsets[id];
            memcpy(dst, &f, 4);
        }
        for (uint32_t i = lane; i < actual_dim; i += 64) dst[4 + i] = codes[(uint64_t)id * actual_dim + i];
    }
}

int32_t launch_sq_split(hipStream_t st, const void *rows, uint64_t row_stride, uint64_t n, uint32_t actual_dim, void *codes,
                        float *offsets) {
    if (n == 0) return QMX_OK;
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(sq_split_kernel, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, st, (const uint8_t *)rows, row_stride, n,
                       actual_dim, (uint8_t *)codes, offsets, 0, nullptr, 0, nullptr);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}
int32_t launch_sq_gather_rows(hipStream_t st, const void *codes, const float *offsets, uint32_t actual_dim, const uint32_t *ids,
                              uint32_t n, uint64_t n_rows, void *rows_out, int *err_flag) {
    if (n == 0) return QMX_OK;
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(sq_split_kernel, dim3((n + 3) / 4), dim3(256), 0, st, (const uint8_t *)rows_out, 4 + (uint64_t)actual_dim,
                       (uint64_t)n, actual_dim, (uint8_t *)const_cast<void *>(codes), const_cast<float *>(offsets), 1, ids, n_rows,
                       err_flag);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

// FilteredScorer::new_internal for SQ (encode_internal_vector :715-728): stored row i becomes the
// query: codes copied, query offset = vector_offset_i - shift (postprocess_internal_score :105-114)
__global__ __launch_bounds__(64) void sq_internal_query_kernel(const uint8_t *codes, const float *offsets, uint32_t actual_dim,
                                                               const uint32_t *ids, uint64_t n_rows, float shift,
                                                               uint8_t *tile, uint32_t q_stride, uint32_t aux_off, int *err_flag) {
    const uint32_t q = blockIdx.x;
    const int lane = threadIdx.x;
    const uint32_t id = ids[q];
    uint8_t *dst = tile + (uint64_t)q * q_stride;
    for (uint32_t i = lane; i < q_stride; i += 64) dst[i] = 0;
    __syncthreads();
    if (id >= n_rows) {
        if (lane == 0) *err_flag = 1;
        return;
    }
    for (uint32_t i = lane; i < actual_dim; i += 64) dst[i] = codes[(uint64_t)id * actual_dim + i];
    if (lane == 0) reinterpret_cast<QueryAux *>(dst + aux_off)->f0 = offsets[id] - shift;
}
int32_t launch_sq_internal_query(hipStream_t st, const void *codes, const float *offsets, uint32_t actual_dim, const uint32_t *ids,
                                 uint32_t nq, uint64_t n_rows, float shift, void *tile, uint32_t q_stride, uint32_t aux_off,
                                 int *err_flag) {
    if (nq == 0) return QMX_OK;
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(sq_internal_query_kernel, dim3(nq), dim3(64), 0, st, (const uint8_t *)codes, offsets, actual_dim, ids,
                       n_rows, shift, (uint8_t *)tile, q_stride, aux_off, err_flag);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

}  // namespace qmx

namespace qmx {
static_assert(HopRow<RowSQInternal<false, false>>::INTERNAL_QOFF && HopRow<RowSQInternal<true, false>>::INTERNAL_QOFF &&
                  !HopRow<RowSQ<false, false>>::INTERNAL_QOFF,
              "only the build-time SQ policy takes its query offset from ScanArgs::sq_qoff");
}

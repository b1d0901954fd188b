// scan_tq.hip — EncodedVectorsTQ (TurboQuant, lib/quantization/src/turboquant/ behind lib/quantization/src/encoded_vectors_tq.rs) on device.
//
// Reference:
//   TurboQuantizer::precompute_query   turboquant/quantization.rs:496-567   rotate the (preprocessed) query in f64, narrow to f32, encode as integers
//   HadamardRotation::apply            turboquant/rotation.rs:69-131,264-280 WHT + 1/sqrt(size) over decreasing power-of-two chunks, three
//                                                                             fixed permutations (permutation.rs: Knuth-MMIX LCG Fisher-Yates)
//   Query{4,2}bitSimd::{new, dotprod}  turboquant/simd/query{4,2}bit/mod.rs   q_signed = clamp(round(v * 8127 / max|v|)); dot_raw = sum q_signed * c_u;
//                                                                             score = postprocess_scale * (dot_raw - 128 sum q_signed) as f32
//   Query1bitSimd<8>::{new, dotprod}   turboquant/simd/query1bit/mod.rs       q in [-127, 127] as 8 two's-complement bit planes per 16-byte block;
//                                                                             v.q = sum_b w_b popcount(v & plane_b), score = scale * (2 v.q - sum q)
//   score_precomputed                  turboquant/quantization.rs:569-620     dot * scaling_factor | |q|^2 + |v|^2 - 2 dot scaling_factor  (then `invert`)
//   score_symmetric                    turboquant/quantization.rs:395-445 + score_{4,2,1}bit_internal (simd/*/mod.rs)
//   row layout                         turboquant/encoding.rs:117-134,172-258 [codes: padded_dim * bits / 8 bytes, LSB first][scaling_factor f32][l2 f32 (L2)]
// The x86_64 constants are the ones restated (the reference's aarch64 build uses other integer ranges: its scores differ in the last bits).
// Everything between the rotation and the final f32 products is integer arithmetic, exact in any order: scores are bit-identical to the
// reference's whatever SIMD path it took.  TQMode::Normal and TQMode::Plus (error correction given or fitted by tq_p2_fit_kernel below); distances Dot, Cosine, L2.
//
// HBM layout: code block [n][code bytes rounded up to 16] (zero padded) + one or two f32 columns (scaling_factor, l2_length), like SQ.
// A query entry holds QPIECES 16-byte pieces per 16-byte row piece, ordered the way the decode of a row dword produces its operands:
//   4 bits  byte k of a row piece = dims 2k (low nibble), 2k + 1 (high nibble): pieces [low half of the even dims][low half of the odd dims]
//           [high half of the even dims][high half of the odd dims]  (q_signed = 128 high + low, both halves in [-64, 63])
//   2 bits  byte k = dims 4k .. 4k + 3: pieces [low half of dims = j mod 4] j = 0..3, then the four high-half pieces
//   1 bit   bit i of the piece = dim i: pieces = the 8 bit planes of q
#include "tq_policies.hpp"
#include "tq_rotate.hpp"

namespace qmx {

// the scan / pair kernels of each bit width are a translation unit of their own (scan_tq4.hip, scan_tq2.hip, scan_tq1.hip: compile time)
int32_t launch_scan_tq(hipStream_t st, int qt, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    switch (a.tq_bits) {
        case 4: return launch_scan_tq4(st, qt, mode, a, num_cus, grid_out);
        case 2: return launch_scan_tq2(st, qt, mode, a, num_cus, grid_out);
        case 1: return launch_scan_tq1(st, qt, mode, a, num_cus, grid_out);
    }
    set_error("TurboQuant: %u bits per value not supported", a.tq_bits);
    return QMX_ERR_NOT_SUPPORTED;
}
int32_t launch_pairs_tq(hipStream_t st, const ScanArgs &a, const PairSel &sel, uint64_t n_items, int num_cus) {
    switch (a.tq_bits) {
        case 4: return launch_pairs_tq4(st, a, sel, n_items, num_cus);
        case 2: return launch_pairs_tq2(st, a, sel, n_items, num_cus);
        case 1: return launch_pairs_tq1(st, a, sel, n_items, num_cus);
    }
    set_error("TurboQuant: %u bits per value not supported", a.tq_bits);
    return QMX_ERR_NOT_SUPPORTED;
}


// ---- upload: reference rows [codes][extras] -> code block (16-byte multiple, zero padded) + extras columns ----
__global__ __launch_bounds__(256) void tq_split_kernel(const uint8_t *rows, uint64_t src_stride, uint64_t n, uint32_t code_bytes, uint32_t dst_stride,
                                                       int has_l2, uint8_t *codes, float *sf, float *l2, float *xm) {
    const uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const int lane = threadIdx.x & 63;
    const uint8_t *src = rows + r * src_stride;
    uint8_t *dst = codes + r * dst_stride;
    for (uint32_t i = lane; i < dst_stride; i += 64) dst[i] = i < code_bytes ? src[i] : 0;
    if (lane == 0) {
        float f;
        memcpy(&f, src + code_bytes, 4);
        sf[r] = f;
        if (has_l2) {
            memcpy(&f, src + code_bytes + 4, 4);
            l2[r] = f;
        }
        if (xm) {                                                         // TQ+: the trailing f32 of the extras
            memcpy(&f, src + code_bytes + (has_l2 ? 8 : 4), 4);
            xm[r] = f;
        }
    }
}
int32_t launch_tq_split(hipStream_t st, const void *rows, uint64_t src_stride, uint64_t n, uint32_t code_bytes, uint32_t dst_stride, int has_l2,
                        void *codes, float *sf, float *l2, float *xm) {
    if (n == 0) return QMX_OK;
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(tq_split_kernel, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, st, (const uint8_t *)rows, src_stride, n, code_bytes, dst_stride, has_l2,
                       (uint8_t *)codes, sf, l2, xm);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}
// ---- HadamardRotation::apply for a batch of vectors: in [n][dim] f32 -> out [n][padded_dim] f64 (the zero padding past rot_dim untouched) ----
// One block per vector, the vector in LDS (two f64 buffers: the gathers ping-pong).  Every element sees the reference's operation sequence:
// one add or sub per butterfly stage in ascending stride order, one multiply by 1 / sqrt(size), so the f64 results are bit-identical.
__device__ __forceinline__ void tq_wht_chunks(double *x, const TqRotation &r) {
    for (uint32_t c = 0; c < r.n_chunks; ++c) {
        double *xc = x + r.chunk_off[c];
        const uint32_t size = r.chunk_size[c];
        for (uint32_t h = 1; h < size; h *= 2) {
            for (uint32_t p = threadIdx.x; p < size / 2; p += blockDim.x) {
                const uint32_t j = (p / h) * 2 * h + (p % h);
                const double a = xc[j], b = xc[j + h];
                xc[j] = a + b;
                xc[j + h] = a - b;
            }
            __syncthreads();
        }
        const double norm = r.chunk_norm[c];
        for (uint32_t i = threadIdx.x; i < size; i += blockDim.x) xc[i] *= norm;
    }
    __syncthreads();
}
template <class T>      // T = float: vectors as given (HadamardRotation::apply); double: dequantized rows, possibly in place (apply_inverse, with the backward maps)
__global__ __launch_bounds__(256) void tq_rotate_kernel(const T *in, uint64_t in_stride, uint32_t n, TqRotation r, double *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_tq[];
    double *a = reinterpret_cast<double *>(smem_tq), *b = a + r.rot_dim;
    const uint32_t v = blockIdx.x;
    if (v >= n) return;
    const T *src = in + (uint64_t)v * in_stride;
    for (uint32_t i = threadIdx.x; i < r.rot_dim; i += blockDim.x) a[i] = i < r.dim ? (double)src[i] : 0.0;
    __syncthreads();
    tq_wht_chunks(a, r);
    double *s = a, *d = b;
    for (int p = 0; p < 3; ++p) {
        const uint32_t *map = r.maps + (size_t)p * r.rot_dim;
        for (uint32_t k = threadIdx.x; k < r.rot_dim; k += blockDim.x) d[k] = s[map[k]];
        __syncthreads();
        tq_wht_chunks(d, r);
        double *t = s; s = d; d = t;
    }
    double *o = out + (uint64_t)v * r.padded_dim;
    for (uint32_t i = threadIdx.x; i < r.padded_dim; i += blockDim.x) o[i] = i < r.rot_dim ? s[i] : (i < r.dim ? (double)src[i] : 0.0);
}
// The same rotation with ONE VECTOR PER WAVE, for rotations of a multiple of E coordinates (E = 16 / 32 / 64 per lane, up to 64 E in all): element i
// lives in lane i / E, register i % E.  Every power-of-two chunk then covers whole lanes, aligned: butterfly stages of stride < E run inside the lane,
// stages of stride h >= E pair lane l with l ^ (h / E) (a lane joins while its chunk is longer than h) - the same adds and subtracts per element in the
// same ascending-stride order, then the chunk's norm: the same bits as the block kernel above, without its 40-odd block barriers per vector.  The
// permutation between two transforms goes through a per-wave LDS image of the vector.
template <class T, int E>
__global__ __launch_bounds__(64) void tq_rotate_wave_kernel(const T *in, uint64_t in_stride, uint32_t n, TqRotation r, double *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_tqw[];
    double *buf = reinterpret_cast<double *>(smem_tqw);
    const int lane = threadIdx.x;
    const uint32_t v = blockIdx.x;
    if (v >= n) return;
    const T *src = in + (uint64_t)v * in_stride;
    const uint32_t first = (uint32_t)lane * E;                 // this lane's first coordinate
    const bool act = first < r.rot_dim;
    uint32_t my_size;
    double my_norm;
    tq_wave_lane_chunk<E>(r, lane, &my_size, &my_norm);
    double x[E];
#pragma unroll
    for (int k = 0; k < E; ++k) x[k] = (act && first + k < r.dim) ? (double)src[first + k] : 0.0;
    tq_wave_rotate<E>(x, r, buf, my_size, my_norm, lane);
    double *o = out + (uint64_t)v * r.padded_dim;
    // (in place - the inverse rotation - too: the wave read its whole vector before this point, and a coordinate past the rotation is read and written
    // by the same lane)
    if (act) {
#pragma unroll
        for (int k = 0; k < E; ++k) o[first + k] = x[k];
    }
    for (uint32_t i = r.rot_dim + (uint32_t)lane; i < r.padded_dim; i += 64) o[i] = i < r.dim ? (double)src[i] : 0.0;
}
template <class T>
static int32_t launch_tq_rotate_any(hipStream_t st, const T *d_in, uint64_t in_stride, uint32_t n, const TqRotation &r, double *d_out) {
    const uint32_t rd = r.rot_dim;
    int e = 0;
    if (!option(OPT_TQ_ROTATE_BLOCK) && rd >= 64) {
        if (rd % 16 == 0 && rd <= 1024) e = 16;
        else if (rd % 32 == 0 && rd <= 2048) e = 32;
        else if (rd % 64 == 0 && rd <= 4096) e = 64;
    }
    ::qmx::clear_stale_error();
    const size_t lds_w = (size_t)rd * sizeof(double);
    if (e == 16) hipLaunchKernelGGL((tq_rotate_wave_kernel<T, 16>), dim3(n), dim3(64), lds_w, st, d_in, in_stride, n, r, d_out);
    else if (e == 32) hipLaunchKernelGGL((tq_rotate_wave_kernel<T, 32>), dim3(n), dim3(64), lds_w, st, d_in, in_stride, n, r, d_out);
    else if (e == 64) hipLaunchKernelGGL((tq_rotate_wave_kernel<T, 64>), dim3(n), dim3(64), lds_w, st, d_in, in_stride, n, r, d_out);
    else {
        const size_t lds = (size_t)2 * rd * sizeof(double);
        QMX_REQUIRE(lds <= 150 * 1024, QMX_ERR_NOT_SUPPORTED, "TurboQuant rotation over %u coordinates does not fit the LDS", rd);
        static thread_local DeviceOnce attr_once;
        if (attr_once.need()) {
            QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(tq_rotate_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(tq_rotate_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            attr_once.mark();
        }
        hipLaunchKernelGGL(tq_rotate_kernel<T>, dim3(n), dim3(256), lds, st, d_in, in_stride, n, r, d_out);
    }
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}
int32_t launch_tq_rotate(hipStream_t st, const float *d_in, uint32_t n, const TqRotationHost &h, double *d_out) {
    if (n == 0) return QMX_OK;
    TqRotation r;
    r.maps = h.d_maps; r.chunk_off = h.d_chunk_off; r.chunk_size = h.d_chunk_size; r.chunk_norm = h.d_chunk_norm;
    r.n_chunks = h.n_chunks; r.rot_dim = h.rot_dim; r.padded_dim = h.padded_dim; r.dim = h.dim;
    return launch_tq_rotate_any<float>(st, d_in, (uint64_t)h.dim, n, r, d_out);
}
// HadamardRotation::apply_inverse on [n][padded_dim] f64 vectors in place: `h` carries the backward maps, last permutation first (api_*.hip
// tq_rotation_inverse); the coordinates past rot_dim stay as they are (quantization.rs:382-388)
int32_t launch_tq_rotate_f64(hipStream_t st, double *d_buf, uint32_t n, const TqRotationHost &h) {
    if (n == 0) return QMX_OK;
    TqRotation r;
    r.maps = h.d_maps; r.chunk_off = h.d_chunk_off; r.chunk_size = h.d_chunk_size; r.chunk_norm = h.d_chunk_norm;
    r.n_chunks = h.n_chunks; r.rot_dim = h.rot_dim; r.padded_dim = h.padded_dim; r.dim = h.padded_dim;
    return launch_tq_rotate_any<double>(st, (const double *)d_buf, (uint64_t)h.padded_dim, n, r, d_buf);
}

// ---- TurboQuantizer::quantize on rotated vectors: rot [n][padded_dim] f64 -> reference rows ----
// The f64 sums of the reference (l2 length, <X, M>, the degenerate test, the centroid norm) are iterator sums: one add per element, in index order - a
// chain nobody can split without changing bits.  So the kernel runs ONE VECTOR PER LANE: a wave takes 64 vectors, walks them 16 elements at a time
// (the 64 x 16 doubles arrive coalesced, 128 bytes per vector, and are transposed through LDS), and every lane carries its own vector's chains; the
// elementwise work (rescale, TQ+ correction, nearest centroid, bit packing) rides along.  Passes over the data: the l2 length (dot / euclid), the TQ+
// sums (TQ+ only), then codes + norms.  (The first version ran one block per vector with the chains in thread 0: 5.25 s for 10 M x 768.)
constexpr int TQQ_CH = 16;                  // elements per trip
__global__ __launch_bounds__(64) void tq_quantize_kernel(const double *rot, uint32_t n, uint32_t padded_dim, uint32_t value_bits, uint32_t distance, uint8_t *out,
                                                         uint32_t out_stride, const float *shift, const float *scale) {
    __shared__ double tile[64][TQQ_CH + 1];
    __shared__ float cent[16];
    const int lane = threadIdx.x;
    const uint64_t v0 = (uint64_t)blockIdx.x * 64;
    const bool live = v0 + lane < n;
    const bool has_l2 = distance != QMX_DISTANCE_COSINE;
    const uint32_t code_bytes = padded_dim * value_bits / 8;
    // centroids and the midpoint boundaries of lloyd_max.rs: index = boundaries.partition_point(|&b| (val as f32) > b)
    const float C1[2] = {-0.7978846f, 0.7978846f};
    const float C2[4] = {-1.510f, -0.4528f, 0.4528f, 1.510f};
    const float C4[16] = {-2.733f, -2.069f, -1.618f, -1.256f, -0.9424f, -0.6568f, -0.3881f, -0.1284f, 0.1284f, 0.3881f, 0.6568f, 0.9424f, 1.256f, 1.618f, 2.069f, 2.733f};
    if (lane < 16) cent[lane] = value_bits == 4 ? C4[lane] : value_bits == 2 ? C2[lane & 3] : C1[lane & 1];
    // elements [c0, c0 + 16) of the wave's 64 vectors -> tile (zeros past the vector / past n): lane l fetches doubles 2 (l % 8), + 1 of vector 8 t + l / 8
    auto load_chunk = [&](uint32_t c0) {
        __syncthreads();                    // the previous trip's reads are done
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int vec = 8 * t + (lane >> 3), e = 2 * (lane & 7);
            double a = 0.0, b = 0.0;
            if (v0 + vec < n && c0 + e < padded_dim) {
                const double *src = rot + (v0 + vec) * padded_dim + c0 + e;
                a = src[0];
                b = src[1];                 // (padded_dim is even: a vector never ends between the two)
            }
            tile[vec][e] = a;
            tile[vec][e + 1] = b;
        }
        __syncthreads();
    };
    auto centroid_index = [&](double val) -> uint32_t {          // the boundaries ascend: the partition point is the number of boundaries below the value
        const float f = (float)val;
        uint32_t idx = 0;
        if (value_bits == 4) {
#pragma unroll
            for (int i = 0; i < 15; ++i) idx += f > (C4[i] + C4[i + 1]) / 2.0f ? 1u : 0u;
        } else if (value_bits == 2) {
#pragma unroll
            for (int i = 0; i < 3; ++i) idx += f > (C2[i] + C2[i + 1]) / 2.0f ? 1u : 0u;
        } else {
            idx = f > (C1[0] + C1[1]) / 2.0f ? 1u : 0u;
        }
        return idx;
    };
    // ---- pass 1: l2 length (turboquant/quantization.rs:169-207) ----
    float l2_length = 1.0f;
    if (has_l2) {
        double s = 0.0;
        for (uint32_t c0 = 0; c0 < padded_dim; c0 += TQQ_CH) {
            load_chunk(c0);
            const uint32_t cnt = padded_dim - c0 < TQQ_CH ? padded_dim - c0 : TQQ_CH;
            for (uint32_t e = 0; e < cnt; ++e) {
                const double x = tile[lane][e];
                s = s + x * x;
            }
        }
        l2_length = (float)sqrt(s);
    }
    const double length = (double)l2_length;
    const bool rescale = length > 0.0;
    const double length_scale = rescale ? sqrt((double)padded_dim) / length : 1.0;
    // ---- pass 2 (TQ+, :231-247): xm = <X, M> on the rescaled vector, then x <- (x + shift) * scale; skipped for an all-zero vector ----
    bool apply = false;
    float xm_f = 0.0f;
    if (shift) {
        double l2sq = 0.0, xm = 0.0;
        for (uint32_t c0 = 0; c0 < padded_dim; c0 += TQQ_CH) {
            load_chunk(c0);
            const uint32_t cnt = padded_dim - c0 < TQQ_CH ? padded_dim - c0 : TQQ_CH;
            for (uint32_t e = 0; e < cnt; ++e) {
                double x = tile[lane][e];
                if (rescale) x = x * length_scale;
                l2sq = l2sq + x * x;
                xm = xm + x * (double)(-shift[c0 + e]);
            }
        }
        apply = !(l2sq < 1e-12);
        xm_f = apply ? (float)xm : 0.0f;
    }
    // ---- pass 3: codes (BitWriter, LSB first), the degenerate test of cosine rows, the centroid norm ----
    uint8_t *row = out + (v0 + lane) * out_stride;
    const uint32_t per = 8 / value_bits;
    double s_deg = 0.0, sq = 0.0;
    for (uint32_t c0 = 0; c0 < padded_dim; c0 += TQQ_CH) {
        load_chunk(c0);
        const uint32_t cnt = padded_dim - c0 < TQQ_CH ? padded_dim - c0 : TQQ_CH;
        uint64_t bits = 0;
        for (uint32_t e = 0; e < cnt; ++e) {
            double x = tile[lane][e];
            if (rescale) x = x * length_scale;
            if (apply) x = (x + (double)shift[c0 + e]) * (double)scale[c0 + e];
            const uint32_t idx = centroid_index(x);
            bits |= (uint64_t)idx << (e * value_bits);
            s_deg = s_deg + x * x;
            double c = (double)cent[idx];
            if (shift) c = c / (double)scale[c0 + e] - (double)shift[c0 + e];      // compute_centroid_norm reverts the correction (:311-314)
            sq = sq + c * c;
        }
        if (live) {
            const uint32_t nb = cnt / per;                       // whole bytes: padded_dim is a multiple of `per`
            uint8_t *dst = row + (uint64_t)c0 * value_bits / 8;
            if (nb == 8 && (reinterpret_cast<uintptr_t>(dst) & 3u) == 0) {
                reinterpret_cast<uint32_t *>(dst)[0] = (uint32_t)bits;
                reinterpret_cast<uint32_t *>(dst)[1] = (uint32_t)(bits >> 32);
            } else if (nb == 4 && (reinterpret_cast<uintptr_t>(dst) & 3u) == 0) {
                reinterpret_cast<uint32_t *>(dst)[0] = (uint32_t)bits;
            } else {
                for (uint32_t b = 0; b < nb; ++b) dst[b] = (uint8_t)(bits >> (8 * b));
            }
        }
    }
    if (!live) return;
    const bool degenerate = distance == QMX_DISTANCE_COSINE && s_deg < 1e-12;
    const float centroid_norm = degenerate ? sqrtf((float)padded_dim) : (float)sqrt(sq);
    // pack_extras_into (encoding.rs:218-247): l2 / centroid norm; DistanceType::L1 stores the bare l2 length
    const float scaling_factor = distance == QMX_DISTANCE_MANHATTAN ? l2_length : (has_l2 ? l2_length : 1.0f) / centroid_norm;
    memcpy(row + code_bytes, &scaling_factor, 4);
    if (distance == QMX_DISTANCE_EUCLID) memcpy(row + code_bytes + 4, &l2_length, 4);
    if (shift) memcpy(row + code_bytes + (distance == QMX_DISTANCE_EUCLID ? 8 : 4), &xm_f, 4);
}
// ---- TQ+ parameter fit: the first pass of EncodedVectorsTQ::encode (encoded_vectors_tq.rs:156-234) ----
// preprocess_into's length rescale (turboquant/quantization.rs:169-207) of already rotated vectors, in place: norm -> sqrt(padded_dim)
__global__ __launch_bounds__(256) void tq_rescale_kernel(double *rot, uint32_t n, uint32_t padded_dim, uint32_t distance) {
    __shared__ double sh_scale;
    __shared__ int sh_apply;
    const uint32_t v = blockIdx.x;
    if (v >= n) return;
    double *x = rot + (uint64_t)v * padded_dim;
    if (threadIdx.x == 0) {
        float l2_length = 1.0f;
        if (distance != QMX_DISTANCE_COSINE) {
            double s = 0.0;
            for (uint32_t i = 0; i < padded_dim; ++i) s = s + x[i] * x[i];
            l2_length = (float)sqrt(s);
        }
        const double length = (double)l2_length;
        sh_apply = length > 0.0;
        sh_scale = length > 0.0 ? sqrt((double)padded_dim) / length : 1.0;
    }
    __syncthreads();
    if (sh_apply)
        for (uint32_t i = threadIdx.x; i < padded_dim; i += blockDim.x) x[i] = x[i] * sh_scale;
}
// The extended P-square estimator (p_square.rs; Jain & Chlamtac 1985 with P2_MARKERS = 7 markers), x86_64 AVX2 + FMA dispatch: the desired
// positions of markers 0..3 are one fused multiply-add, markers 4..6 `1.0 + p * (count - 1)`; find_marker counts heights[1..N-1] < x.
struct P2 {
    static constexpr int N = 7;
    int count;
    double q, h[N], n[N], nd[N], tp[N];     // before the N-th observation h[] holds the observations
    __device__ void init(double quantile) {
        count = 0;
        q = quantile;
        const int extra = (N - 5) / 2;      // generate_grid_probabilities (:171-214)
        tp[0] = 0.0;
        tp[1] = q * 0.5;
        for (int i = 0; i < extra; ++i) tp[i + 2] = q * (0.7 + 0.3 * (double)(i + 1) / ((double)extra + 2.0));
        tp[N / 2] = q;
        for (int i = 0; i < extra; ++i) tp[N / 2 + 1 + i] = 1.0 + (q - 1.0) * (0.7 + 0.3 * (double)(extra - i) / ((double)extra + 2.0));
        tp[N - 2] = 1.0 + (q - 1.0) * 0.5;
        tp[N - 1] = 1.0;
    }
    __device__ static void sort(double *b, int m) {
        for (int i = 1; i < m; ++i) {
            const double x = b[i];
            int j = i - 1;
            while (j >= 0 && b[j] > x) { b[j + 1] = b[j]; --j; }
            b[j + 1] = x;
        }
    }
    __device__ void step(int i, double dsign) {     // adjust_step (:456-488)
        const double prev_h = h[i - 1], next_h = h[i + 1], prev_n = n[i - 1], next_n = n[i + 1], cur_h = h[i], cur_n = n[i];
        const double denom = next_n - prev_n;
        double h_par = cur_h;
        if (denom != 0.0) {
            const double a = (cur_n - prev_n + dsign) / (next_n - cur_n) * (next_h - cur_h);
            const double b = (next_n - cur_n - dsign) / (cur_n - prev_n) * (cur_h - prev_h);
            h_par = cur_h + (a + b) * dsign / denom;
        }
        if (h_par > prev_h && h_par < next_h && isfinite(h_par)) h[i] = h_par;
        else if (dsign > 0.0) h[i] = cur_h + (next_h - cur_h) / (next_n - cur_n);
        else h[i] = cur_h + (prev_h - cur_h) / (prev_n - cur_n);
        n[i] += dsign;
    }
    __device__ void push(double x) {
        if (isnan(x) || !isfinite(x)) return;
        if (count < N) {
            h[count++] = x;
            if (count == N) {                   // new_from_linear (:91-116)
                sort(h, N);
                for (int i = 0; i < N; ++i) {
                    n[i] = (double)(i + 1);
                    nd[i] = 1.0 + tp[i] * (double)(N - 1);
                }
            }
            return;
        }
        count += 1;
        int k;
        if (x < h[0]) { h[0] = x; k = 0; }
        else if (x > h[N - 1]) { h[N - 1] = x; k = N - 1; }
        else { k = 0; for (int i = 1; i < N; ++i) k += h[i] < x ? 1 : 0; }
        for (int i = 0; i < N; ++i) n[i] += i > k ? 1.0 : 0.0;
        const double cm1 = (double)(count - 1);
        for (int i = 0; i < N; ++i) nd[i] = i < (N / 4) * 4 ? fma(tp[i], cm1, 1.0) : 1.0 + tp[i] * cm1;
        for (int i = 1; i < N - 1; ++i) {       // adjust_marker (:438-454)
            for (;;) {
                const double di = nd[i] - n[i];
                if (di >= 1.0 && (n[i + 1] - n[i]) > 1.0) step(i, 1.0);
                else if (di <= -1.0 && (n[i - 1] - n[i]) < -1.0) step(i, -1.0);
                else break;
            }
        }
    }
    __device__ double estimate() {              // :61-66, :118-121, estimate_quantile_from_slice (:502-521)
        if (count >= N) return h[N / 2];
        if (count == 0) return 0.0;
        if (count == 1) return h[0];
        sort(h, count);
        const double k = q * ((double)count - 1.0);
        const int lo = (int)floor(k), hi = (int)ceil(k);
        if (lo == hi) return h[lo];
        return h[lo] + (k - (double)lo) * (h[hi] - h[lo]);
    }
};
// one thread per rotated coordinate: both estimators over the sample in order (find_quantile_interval_per_coordinate_with_preprocess,
// quantile.rs:130-281), then shift / scale (encoded_vectors_tq.rs:219-233)
__global__ __launch_bounds__(64) void tq_p2_fit_kernel(const double *rot, uint32_t n, uint32_t padded_dim, double min_q, double max_q, float c_outer,
                                                       float *shift, float *scale) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= padded_dim) return;
    P2 lo, hi;
    lo.init(min_q);
    hi.init(max_q);
    for (uint32_t v = 0; v < n; ++v) {
        const double x = rot[(uint64_t)v * padded_dim + d];
        lo.push(x);
        hi.push(x);
    }
    float q_lo = 0.0f, q_hi = 0.0f;
    if (n) { q_lo = (float)lo.estimate(); q_hi = (float)hi.estimate(); }
    shift[d] = -(q_lo + q_hi) / 2.0f;
    const float denom = q_hi - q_lo;
    scale[d] = denom > 1e-3f ? (2.0f * c_outer) / denom : 1.0f;     // MIN_QUANTILE_WIDTH
}
int32_t launch_tq_plus_fit(hipStream_t st, double *d_rot, uint32_t n, uint32_t padded_dim, uint32_t distance, double min_q, double max_q, float c_outer,
                           float *d_shift, float *d_scale) {
    ::qmx::clear_stale_error();
    if (n) hipLaunchKernelGGL(tq_rescale_kernel, dim3(n), dim3(256), 0, st, d_rot, n, padded_dim, distance);
    hipLaunchKernelGGL(tq_p2_fit_kernel, dim3((padded_dim + 63) / 64), dim3(64), 0, st, (const double *)d_rot, n, padded_dim, min_q, max_q, c_outer, d_shift,
                       d_scale);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

int32_t launch_tq_quantize(hipStream_t st, double *d_rot, uint32_t n, uint32_t padded_dim, uint32_t value_bits, uint32_t distance, void *d_out,
                           uint32_t out_stride, const float *d_shift, const float *d_scale) {
    if (n == 0) return QMX_OK;
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(tq_quantize_kernel, dim3((n + 63) / 64), dim3(64), 0, st, d_rot, n, padded_dim, value_bits, distance, (uint8_t *)d_out, out_stride, d_shift, d_scale);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

// ---- Query{N}bitSimd::new on the rotated queries: rot [nq][padded_dim] f64 -> tile entries ----
__global__ __launch_bounds__(256) void tq_query_encode_kernel(double *rot, uint32_t padded_dim, uint32_t bits, int need_l2, uint8_t *tile,
                                                              uint32_t q_stride, uint32_t aux_off, const float *shift, const float *scale, uint32_t qbytes_off) {
    __shared__ float sh_max[256];
    __shared__ float sh_scale;
    __shared__ unsigned long long sh_sum;
    __shared__ float sh_l2q, sh_qm;
    const uint32_t q = blockIdx.x;
    double *x = rot + (uint64_t)q * padded_dim;
    const uint32_t planes = (bits == 1 && shift) ? 16u : 8u;           // TQ+ over 1-bit storage: Query1bitSimd<16> (:557-563)
    // the query's l2 norm (of the rotated query, before the TQ+ rescale) and, under TQ+, qm = <Q, M>, then Q .* D' (:526-540): f64, in order
    if (threadIdx.x == 0) {
        float l2 = 1.0f;
        if (need_l2) {                                                             // rotated.iter().map(|&i| i * i).sum::<f64>().sqrt() as f32
            double s = 0.0;
            for (uint32_t i = 0; i < padded_dim; ++i) s = s + x[i] * x[i];
            l2 = (float)sqrt(s);
        }
        sh_l2q = l2;
        double qm = 0.0;
        if (shift)
            for (uint32_t i = 0; i < padded_dim; ++i) qm = qm + x[i] * (double)(-shift[i]);
        sh_qm = (float)qm;
    }
    __syncthreads();
    if (shift) {
        for (uint32_t i = threadIdx.x; i < padded_dim; i += blockDim.x) x[i] = x[i] / (double)scale[i];
        __syncthreads();
    }
    uint8_t *entry = tile + (uint64_t)q * q_stride;
    // zero the entry's pieces (dims past padded_dim inside the last 16-byte row piece must read as q = 0)
    for (uint32_t i = threadIdx.x; i < aux_off / 4; i += blockDim.x) reinterpret_cast<uint32_t *>(entry)[i] = 0;
    float m = 0.0f;
    for (uint32_t i = threadIdx.x; i < padded_dim; i += blockDim.x) m = fmaxf(m, fabsf((float)x[i]));
    sh_max[threadIdx.x] = m;
    if (threadIdx.x == 0) sh_sum = 0;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) sh_max[threadIdx.x] = fmaxf(sh_max[threadIdx.x], sh_max[threadIdx.x + o]);
        __syncthreads();
    }
    const float abs_max_int = bits == 1 ? (planes == 16 ? 32767.0f : 127.0f) : 8127.0f;
    if (threadIdx.x == 0) {
        float q_abs_max = sh_max[0];
        if (!(q_abs_max > 1.1920929e-7f)) q_abs_max = 1.1920929e-7f;              // .max(f32::EPSILON)
        const float q_scale = abs_max_int / q_abs_max;
        sh_scale = q_scale;
        QueryAux *aux = reinterpret_cast<QueryAux *>(entry + aux_off);
        const float codebook_scale = 128.0f / (bits == 4 ? 2.733f : 1.510f);
        aux->f0 = bits == 1 ? 0.7978846f / q_scale : 1.0f / (q_scale * codebook_scale);
        aux->pad[0] = __float_as_uint(sh_l2q);
        aux->pad[3] = __float_as_uint(shift ? sh_qm : 0.0f);                       // EncodedQueryTQ::ec_correction
    }
    __syncthreads();
    const float q_scale = sh_scale;
    long long local = 0;
    for (uint32_t i = threadIdx.x; i < padded_dim; i += blockDim.x) {
        float v = roundf((float)x[i] * q_scale);
        v = fminf(fmaxf(v, -abs_max_int), abs_max_int);
        const int32_t qs = (int32_t)v;
        local += qs;
        if (bits == 1) {
            // planes of the 16-byte block holding dim i: plane b at piece (block * 8 + b), bit (i % 128) of it
            const uint32_t block = i / 128, bit = i % 128;
            const uint32_t u = (uint32_t)qs & (planes == 16 ? 0xFFFFu : 0xFFu);
            for (uint32_t b = 0; b < planes; ++b)
                if ((u >> b) & 1u) atomicOr(reinterpret_cast<uint32_t *>(entry + ((size_t)block * planes + b) * 16 + (bit / 32) * 4), 1u << (bit % 32));
            // the same value as i8 bytes for the matrix-core scan (scan_sq_mfma.hip Tq1Ops): per 16-byte row piece 8 pieces, piece j = the dims
            // = j mod 8 in row-byte order; 16-bit values (TQ+) as q = 256 high + (low - 128), both stored halves in [-128, 127]
            {
                const uint32_t j = bit % 8, k = bit / 8;
                if (planes == 16) {
                    const int32_t hi = qs >> 8, lo = (qs & 255) - 128;   // qs = 256 hi + lo + 128: the scan adds 128 per set bit of the row
                    entry[qbytes_off + ((size_t)block * 16 + j) * 16 + k] = (uint8_t)(int8_t)lo;
                    entry[qbytes_off + ((size_t)block * 16 + 8 + j) * 16 + k] = (uint8_t)(int8_t)hi;
                } else {
                    entry[qbytes_off + ((size_t)byte_form_slot(block) * 8 + j) * 16 + k] = (uint8_t)(int8_t)qs;
                }
            }
        } else {
            // balanced split q_signed = 128 h + l, l in [-64, 63]
            int32_t l_mod = qs % 128;
            if (l_mod < 0) l_mod += 128;                                           // rem_euclid
            const int32_t l = l_mod >= 64 ? l_mod - 128 : l_mod;
            const int32_t hh = (qs - l) / 128;
            if (bits == 4) {
                const uint32_t piece = i / 32, d = i % 32;                         // 32 dims per 16-byte row piece
                const uint32_t odd = d & 1u, k = d >> 1;                           // byte k of the piece, low / high nibble
                entry[((size_t)piece * 4 + odd) * 16 + k] = (uint8_t)(int8_t)l;
                entry[((size_t)piece * 4 + 2 + odd) * 16 + k] = (uint8_t)(int8_t)hh;
            } else {
                const uint32_t piece = i / 64, d = i % 64;                         // 64 dims per row piece
                const uint32_t j = d & 3u, k = d >> 2;
                entry[((size_t)piece * 8 + j) * 16 + k] = (uint8_t)(int8_t)l;
                entry[((size_t)piece * 8 + 4 + j) * 16 + k] = (uint8_t)(int8_t)hh;
            }
        }
    }
    atomicAdd(&sh_sum, (unsigned long long)local);
    __syncthreads();
    if (threadIdx.x == 0) {
        QueryAux *aux = reinterpret_cast<QueryAux *>(entry + aux_off);
        aux->pad[1] = (uint32_t)sh_sum;
        aux->pad[2] = (uint32_t)(sh_sum >> 32);
    }
}
int32_t launch_tq_query_encode(hipStream_t st, double *d_rot, uint32_t nq, uint32_t padded_dim, uint32_t bits, int need_l2, void *tile, uint32_t q_stride,
                               uint32_t aux_off, const float *d_shift, const float *d_scale, uint32_t qbytes_off) {
    if (nq == 0) return QMX_OK;
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(tq_query_encode_kernel, dim3(nq), dim3(256), 0, st, d_rot, padded_dim, bits, need_l2, (uint8_t *)tile, q_stride, aux_off, d_shift, d_scale, qbytes_off);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

// ---- score_symmetric (EncodedVectorsTQ::score_internal): one thread per pair ----
__device__ __constant__ int8_t TQ4_SIGNED[16] = {-128, -97, -76, -59, -44, -31, -18, -6, 6, 18, 31, 44, 59, 76, 97, 127};
__device__ __constant__ int8_t TQ2_SIGNED[4] = {-128, -38, 38, 127};
__global__ __launch_bounds__(256) void tq_internal_kernel(const uint8_t *codes, uint32_t stride, const float *sf, const float *l2, uint32_t code_bytes,
                                                          uint32_t bits, int invert, uint64_t n_rows, const uint32_t *a_ids, const uint32_t *b_ids, uint32_t n,
                                                          float *out, int *err_flag, TqEc ec) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t ia = a_ids[i], ib = b_ids[i];
    if (ia >= n_rows || ib >= n_rows) {
        *err_flag = 1;
        return;
    }
    const uint8_t *a = codes + (uint64_t)ia * stride, *b = codes + (uint64_t)ib * stride;
    float raw_dot;
    if (bits == 1) {
        uint64_t popcnt = 0;
        for (uint32_t k = 0; k < code_bytes; ++k) popcnt += (uint64_t)__popc((uint32_t)(a[k] ^ b[k]));
        const int64_t sign_sum = (int64_t)code_bytes * 8 - 2 * (int64_t)popcnt;
        const float centroid_sq = 0.7978846f * 0.7978846f;
        raw_dot = centroid_sq * (float)sign_sum;
    } else if (ec.weights) {   // score_symmetric_ec (:447-494): sum c_a c_b w_i16 / (weight_scale * CODEBOOK_SCALE^2) + xm_a + xm_b - <M, M>
        int64_t acc = 0;
        if (bits == 4) {
            for (uint32_t k = 0; k < code_bytes; ++k)
                acc += (int64_t)TQ4_SIGNED[a[k] & 15] * TQ4_SIGNED[b[k] & 15] * ec.weights[2 * k] +
                       (int64_t)TQ4_SIGNED[a[k] >> 4] * TQ4_SIGNED[b[k] >> 4] * ec.weights[2 * k + 1];
        } else {
            for (uint32_t k = 0; k < code_bytes; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc += (int64_t)TQ2_SIGNED[(a[k] >> (2 * j)) & 3] * TQ2_SIGNED[(b[k] >> (2 * j)) & 3] * ec.weights[4 * k + j];
        }
        const float codebook_scale = 128.0f / (bits == 4 ? 2.733f : 1.510f);
        const float codebook_scale_sq = codebook_scale * codebook_scale;
        const float weighted = (float)acc / (ec.weight_scale * codebook_scale_sq);
        raw_dot = ((weighted + ec.xm[ia]) + ec.xm[ib]) - ec.mm_const;
    } else {
        int64_t acc = 0;
        if (bits == 4) {
            for (uint32_t k = 0; k < code_bytes; ++k)
                acc += (int64_t)TQ4_SIGNED[a[k] & 15] * TQ4_SIGNED[b[k] & 15] + (int64_t)TQ4_SIGNED[a[k] >> 4] * TQ4_SIGNED[b[k] >> 4];
        } else {
            for (uint32_t k = 0; k < code_bytes; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc += (int64_t)TQ2_SIGNED[(a[k] >> (2 * j)) & 3] * TQ2_SIGNED[(b[k] >> (2 * j)) & 3];
        }
        const float codebook_scale = 128.0f / (bits == 4 ? 2.733f : 1.510f);
        raw_dot = (float)acc / (codebook_scale * codebook_scale);
    }
    const float s1 = sf[ia], s2 = sf[ib];
    float score;
    if (l2) {
        const float x = l2[ia], y = l2[ib];
        score = (x * x + y * y) - ((2.0f * s1) * s2) * raw_dot;
    } else {
        score = (raw_dot * s1) * s2;
    }
    out[i] = invert ? -score : score;
}
int32_t launch_tq_internal(hipStream_t st, const void *codes, uint32_t stride, const float *sf, const float *l2, uint32_t code_bytes, uint32_t bits,
                           int invert, uint64_t n_rows, const uint32_t *a_ids, const uint32_t *b_ids, uint32_t n, float *out, int *err_flag, const TqEc *ec) {
    if (n == 0) return QMX_OK;
    TqEc e{nullptr, nullptr, 1.0f, 0.0f};
    if (ec) e = *ec;
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(tq_internal_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const uint8_t *)codes, stride, sf, l2, code_bytes, bits, invert, n_rows,
                       a_ids, b_ids, n, out, err_flag, e);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

}  // namespace qmx

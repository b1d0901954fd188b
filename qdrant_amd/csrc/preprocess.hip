// preprocess.hip — Metric::preprocess, element casts, row gather.
#include <algorithm>

#include "kernels.hpp"

namespace qmx {

// ------------------------------------------------------------------------------------------
// CosineMetric::preprocess (lib/segment/src/spaces/simple.rs:178-204) with the reference's ISA
// dispatch (simple.rs:15,22): dim >= 32 -> cosine_preprocess_avx (simple_avx.rs:127-165, FMA,
// 4 x 8 accumulators, four_way_hsum), dim >= 16 -> cosine_preprocess_sse (simple_sse.rs:107-150,
// mul+add, 4 x 4 accumulators, hsum128 each then scalar adds), else scalar (simple.rs:228-235).
// One wavefront per vector; lane c owns SIMD accumulator class c and walks i = c, c+C, ... in
// order, so the sum of squares is bit-identical to the x86 reference.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool is_length_zero_or_normalized(float length) {   // spaces/tools.rs:14-16
    return length < 1.1920929e-7f || __builtin_fabsf(length - 1.0f) <= 1.0e-6f;
}

// dim >= 32, dim % 4 == 0, 16-byte aligned rows, 4 rows of a block in 64 KiB of LDS: the row is read ONCE, 16 bytes per lane, into LDS; the 32 accumulator
// classes walk it there (consecutive lanes, consecutive banks) and the division reads it there again - half the global traffic of the kernel below and
// wide loads (10 M x 768: 27 -> 14 ms; a 128-query batch: 18 -> 9 us).  The same chains in the same order: the same bits.
__global__ __launch_bounds__(256) void cosine_preprocess_lds_kernel(const float *in, float *out, uint64_t n, uint32_t dim) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_cp[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t vec = (uint64_t)blockIdx.x * 4 + (uint64_t)wave;
    if (vec >= n) return;
    float *row = reinterpret_cast<float *>(smem_cp) + (size_t)wave * dim;
    const float4 *v4 = reinterpret_cast<const float4 *>(in + vec * dim);
    float4 *o4 = reinterpret_cast<float4 *>(out + vec * dim);
    const uint32_t n4 = dim / 4;
    for (uint32_t i = lane; i < n4; i += 64) reinterpret_cast<float4 *>(row)[i] = v4[i];
    // (one wave writes and reads its own row: LDS operations of a wave complete in order)
    const uint32_t m = dim - dim % 32;
    float acc = 0.0f;
    if (lane < 32)
        for (uint32_t i = lane; i < m; i += 32) acc = __builtin_fmaf(row[i], row[i], acc);
    const float s12 = acc + __shfl_xor(acc, 8, 64);    // sum1 = r0 + r1 | sum2 = r2 + r3
    const float tot = s12 + __shfl_xor(s12, 16, 64);   // total
    const float lr = tot + __shfl_xor(tot, 4, 64);     // hi128 + lo128
    const float pr = lr + __shfl_xor(lr, 1, 64);       // hadd
    float length = pr + __shfl_xor(pr, 2, 64);         // p1 + p2
    length = __shfl(length, 0, 64);
    for (uint32_t i = m; i < dim; ++i) length += row[i] * row[i];
    if (is_length_zero_or_normalized(length)) {
        if (o4 != v4)
            for (uint32_t i = lane; i < n4; i += 64) o4[i] = reinterpret_cast<const float4 *>(row)[i];
        return;
    }
    length = __builtin_sqrtf(length);
    for (uint32_t i = lane; i < n4; i += 64) {
        const float4 x = reinterpret_cast<const float4 *>(row)[i];
        o4[i] = make_float4(x.x / length, x.y / length, x.z / length, x.w / length);   // x / length, not x * (1 / length)
    }
}

__global__ __launch_bounds__(256) void cosine_preprocess_kernel(const float *in, float *out, uint64_t n, uint32_t dim) {
    const int lane = threadIdx.x & 63;
    const uint64_t vec = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (vec >= n) return;
    const float *v = in + vec * dim;
    float *o = out + vec * dim;
    float length;
    if (dim >= 32) {
        const uint32_t m = dim - dim % 32;
        float acc = 0.0f;
        if (lane < 32)
            for (uint32_t i = lane; i < m; i += 32) acc = __builtin_fmaf(v[i], v[i], acc);
        const float s12 = acc + __shfl_xor(acc, 8, 64);    // sum1 = r0 + r1 | sum2 = r2 + r3
        const float tot = s12 + __shfl_xor(s12, 16, 64);   // total
        const float lr = tot + __shfl_xor(tot, 4, 64);     // hi128 + lo128
        const float pr = lr + __shfl_xor(lr, 1, 64);       // hadd
        length = pr + __shfl_xor(pr, 2, 64);               // p1 + p2
        length = __shfl(length, 0, 64);
        for (uint32_t i = m; i < dim; ++i) length += v[i] * v[i];
    } else if (dim >= 16) {
        const uint32_t m = dim - dim % 16;
        float acc = 0.0f;
        if (lane < 16)
            for (uint32_t i = lane; i < m; i += 16) acc = v[i] * v[i] + acc;
        const float x64 = acc + __shfl_xor(acc, 2, 64);    // x + movehl(x)
        const float h = x64 + __shfl_xor(x64, 1, 64);      // add_ss(x64, shuffle 0x55)
        const float h0 = __shfl(h, 0, 64), h1 = __shfl(h, 4, 64), h2 = __shfl(h, 8, 64), h3 = __shfl(h, 12, 64);
        length = ((h0 + h1) + h2) + h3;
        for (uint32_t i = m; i < dim; ++i) length += v[i] * v[i];
    } else {
        length = -0.0f;  // Rust f32 iter::Sum starts from -0.0
        for (uint32_t i = 0; i < dim; ++i) length += v[i] * v[i];
    }
    if (is_length_zero_or_normalized(length)) {
        if (o != v)
            for (uint32_t i = lane; i < dim; i += 64) o[i] = v[i];
        return;
    }
    // IEEE-correct sqrt and divide (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt); NOT
    // __fsqrt_rn / __fdiv_rn: without OCML_BASIC_ROUNDED_OPERATIONS those lower to the 1-ulp native ops
    length = __builtin_sqrtf(length);
    for (uint32_t i = lane; i < dim; i += 64) o[i] = v[i] / length;   // x / length, not x * (1/length)
}

int32_t launch_cosine_preprocess_f32(hipStream_t st, const float *in, float *out, uint64_t n, uint32_t dim) {
    if (n == 0) return QMX_OK;
    const uint32_t blocks = (uint32_t)((n + 3) / 4);
    ::qmx::clear_stale_error();
    if (dim >= 32 && dim % 4 == 0 && (size_t)dim * 16 <= 64 * 1024 && ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0) {
        hipLaunchKernelGGL(cosine_preprocess_lds_kernel, dim3(blocks), dim3(256), (size_t)dim * 16, st, in, out, n, dim);
        QMX_HIP(hipGetLastError());
        return QMX_OK;
    }
    hipLaunchKernelGGL(cosine_preprocess_kernel, dim3(blocks), dim3(256), 0, st, in, out, n, dim);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

// ------------------------------------------------------------------------------------------
// PrimitiveVectorElement::slice_from_float_cow (lib/segment/src/data_types/primitive.rs):
//   f16: half::f16::from_f32, IEEE RNE (:77-79)      u8: `x as u8` saturating truncation, NaN -> 0 (:127-129)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint8_t f32_to_u8_sat(float x);
__global__ void cast_f32_kernel(int dst_dtype, const float *in, void *out, uint64_t count) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const float x = in[i];
        if (dst_dtype == QMX_DTYPE_F16) {
            reinterpret_cast<__half *>(out)[i] = __float2half_rn(x);
        } else if (dst_dtype == QMX_DTYPE_U8) {
            reinterpret_cast<uint8_t *>(out)[i] = f32_to_u8_sat(x);
        } else {
            reinterpret_cast<float *>(out)[i] = x;
        }
    }
}
int32_t launch_cast_f32(hipStream_t st, int dst_dtype, const float *in, void *out, uint64_t count) {
    if (count == 0) return QMX_OK;
    uint64_t blocks = (count + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(cast_f32_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, dst_dtype, in, out, count);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

// ------------------------------------------------------------------------------------------
// Query tile packing = the tail of MetricQueryScorer::new (metric_query_scorer.rs:51-53): cast the
// preprocessed f32 query to the element type (or copy an already-encoded stored row for
// FilteredScorer::new_internal), zero-pad the entry to whole 128-byte segments, fill the aux block.
// One wavefront per query.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint8_t f32_to_u8_sat(float x) {   // `x as u8`: truncating, saturating, NaN -> 0
    return (x != x) ? 0 : (x <= 0.0f ? 0 : (x >= 255.0f ? 255 : (uint8_t)x));
}

__global__ __launch_bounds__(64) void pack_queries_kernel(int dtype, int distance, const void *src, int src_is_encoded,
                                                          uint32_t src_stride, uint32_t dim, unsigned char *tile,
                                                          uint32_t q_stride, uint32_t aux_off) {
    const uint32_t q = blockIdx.x;
    const int lane = threadIdx.x;
    unsigned char *dst = tile + (uint64_t)q * q_stride;
    const unsigned char *sp = reinterpret_cast<const unsigned char *>(src) + (uint64_t)q * src_stride;
    const uint32_t elem = dtype == QMX_DTYPE_F32 ? 4 : dtype == QMX_DTYPE_F16 ? 2 : 1;
    const uint32_t nbytes = dim * elem;
    for (uint32_t i = nbytes + lane; i < q_stride; i += 64) dst[i] = 0;
    if (src_is_encoded) {
        for (uint32_t i = lane; i < nbytes; i += 64) dst[i] = sp[i];
    } else {
        const float *f = reinterpret_cast<const float *>(sp);
        for (uint32_t i = lane; i < dim; i += 64) {
            const float x = f[i];
            if (dtype == QMX_DTYPE_F32) reinterpret_cast<float *>(dst)[i] = x;
            else if (dtype == QMX_DTYPE_F16) reinterpret_cast<__half *>(dst)[i] = __float2half_rn(x);
            else dst[i] = f32_to_u8_sat(x);
        }
    }
    if (dtype == QMX_DTYPE_U8 && distance == QMX_DISTANCE_COSINE && dim >= 32) {
        __syncthreads();
        if (lane == 0) {
            // norm1 of avx_cosine_similarity_bytes (metric_uint/avx2/cosine.rs:47-50,80-99): 8 exact i32
            // lanes (bytes 4j..4j+3 of every 32-byte block), cvtepi32_ps, hsum256, + remainder as f32
            int32_t l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const uint32_t full = dim - dim % 32;
            for (uint32_t i = 0; i < full; ++i) { const int32_t x = dst[i]; l[(i & 31) >> 2] += x * x; }
            int32_t rem = 0;
            for (uint32_t i = full; i < dim; ++i) { const int32_t x = dst[i]; rem += x * x; }
            float lr[4];
            for (int k = 0; k < 4; ++k) lr[k] = (float)l[k + 4] + (float)l[k];
            float n1 = (lr[0] + lr[1]) + (lr[2] + lr[3]);
            if (full < dim) n1 += (float)rem;
            QueryAux *aux = reinterpret_cast<QueryAux *>(dst + aux_off);
            aux->f0 = n1;
            int32_t tot = rem;
            for (int k = 0; k < 8; ++k) tot += l[k];
            aux->i0 = tot;
        }
    }
}

int32_t launch_pack_queries(hipStream_t st, int dtype, int distance, const void *src, int src_is_encoded,
                            uint32_t src_stride, uint32_t nq, uint32_t dim, void *tile, uint32_t q_stride, uint32_t aux_off) {
    if (nq == 0) return QMX_OK;
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(pack_queries_kernel, dim3(nq), dim3(64), 0, st, dtype, distance, src, src_is_encoded, src_stride, dim,
                       (unsigned char *)tile, q_stride, aux_off);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

// ------------------------------------------------------------------------------------------
// row gather: get_dense / get_quantized_vector for a list of ids (one wavefront per row)
// ------------------------------------------------------------------------------------------
__global__ void gather_rows_kernel(const unsigned char *rows, uint64_t row_stride, uint64_t row_bytes,
                                   const uint32_t *ids, uint32_t n, uint64_t n_rows, unsigned char *out, int *err_flag) {
    const int lane = threadIdx.x & 63;
    const uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= n) return;
    const uint32_t id = ids[w];
    if (id >= n_rows) {
        if (lane == 0) *err_flag = 1;
        return;
    }
    const unsigned char *src = rows + (uint64_t)id * row_stride;
    unsigned char *dst = out + (uint64_t)w * row_bytes;
    for (uint64_t i = lane; i < row_bytes; i += 64) dst[i] = src[i];
}
int32_t launch_gather_rows(hipStream_t st, const void *rows, uint64_t row_stride, uint64_t row_bytes,
                           const uint32_t *ids, uint32_t n, uint64_t n_rows, void *out, int *err_flag) {
    if (n == 0) return QMX_OK;
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(gather_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, st, (const unsigned char *)rows,
                       row_stride, row_bytes, ids, n, n_rows, (unsigned char *)out, err_flag);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

// ------------------------------------------------------------------------------------------
// find_min_max_from_iter (lib/quantization/src/quantile.rs:19-33): fold with `value < min` / `value > max` from
// (f32::MAX, f32::MIN): NaN never replaces either.  min / max are order-independent, so a parallel reduction gives
// the reference's result exactly.  out[0] = ordered bits of min, out[1] = ordered bits of max (score_to_ord).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void minmax_kernel(const float *in, uint64_t count, uint32_t *out) {
    uint32_t lo = score_to_ord(3.402823466e+38f), hi = score_to_ord(-3.402823466e+38f);
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256) {
        const float v = in[i];
        if (v == v) {
            const uint32_t o = score_to_ord(v);
            lo = o < lo ? o : lo;
            hi = o > hi ? o : hi;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t l2 = (uint32_t)__shfl_xor((int)lo, off, 64), h2 = (uint32_t)__shfl_xor((int)hi, off, 64);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&out[0], lo);
        atomicMax(&out[1], hi);
    }
}
int32_t launch_minmax_f32(hipStream_t st, const float *in, uint64_t count, float *min_out, float *max_out) {
    uint32_t *d = nullptr;
    QMX_HIP(hipMalloc((void **)&d, 8));
    // host-side images of score_to_ord(f32::MAX) and score_to_ord(f32::MIN)
    const uint32_t init[2] = {0x7F7FFFFFu | 0x80000000u, ~0xFF7FFFFFu};
    hipError_t e = hipMemcpyAsync(d, init, 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && count) {
        ::qmx::clear_stale_error();
        hipLaunchKernelGGL(minmax_kernel, dim3(2048), dim3(256), 0, st, in, count, d);
        e = hipGetLastError();
    }
    uint32_t res[2] = {init[0], init[1]};
    if (e == hipSuccess) e = hipMemcpyAsync(res, d, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    QMX_HIP(e);
    auto unord = [](uint32_t o) {
        union { uint32_t u; float f; } cv;
        cv.u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
        return cv.f;
    };
    *min_out = unord(res[0]);
    *max_out = unord(res[1]);
    return QMX_OK;
}

}  // namespace qmx

// dense_policies.hpp — Metric<f32|f16|u8>::similarity over a device-resident dense block.
//
// Reference leaves restated as lane policies (one policy = one "SIMD leaf" of SURVEY §2.2):
//   f32  lib/segment/src/spaces/simple_avx.rs:32-213        (dot / euclid / manhattan, AVX+FMA order)  bit-exact
//   f16  lib/segment/src/spaces/metric_f16/avx/*.rs          (F16C convert, f32 FMA)                    <= 1e-5
//   u8   lib/segment/src/spaces/metric_uint/avx2/*.rs        (exact i32 lanes, cvtepi32_ps, f32 hsum)   bit-exact
//        lib/segment/src/spaces/metric_uint/simple_*.rs      (scalar order, QMX_SEG_U8_SCALAR_ORDER)   bit-exact
// Compiled with -ffp-contract=off: every fused multiply-add below is an explicit fmaf, every
// separate mul/add stays separate, as in the Rust/C reference.
#pragma once
#include "scan_common.hpp"


namespace qmx {

enum { M_DOT = 0, M_EUCLID = 1, M_MANHATTAN = 2, M_COSINE = 3 };

// ------------------------------------------------------------------------------------------
// f32 : 8 lanes x float4 = the 4 x __m256 accumulators of one 32-float AVX iteration.
// lane (h = t>>2, r = t&3) holds AVX register r, SIMD lanes 4h..4h+3.
// ------------------------------------------------------------------------------------------
template <int METRIC>
struct RowF32 {
    static constexpr int NACC = 4;
    static constexpr int NRAUX = 0;
    static constexpr int R16 = 2;
    typedef float acc_t;

    static __device__ __forceinline__ void mac1(float &a, float q, float v) {
        if (METRIC == M_DOT) {
            a = __builtin_fmaf(q, v, a);                 // _mm256_fmadd_ps(v1, v2, sum)  simple_avx.rs:184
        } else if (METRIC == M_EUCLID) {
            const float d = q - v;                       // _mm256_sub_ps(v1, v2)         simple_avx.rs:48
            a = __builtin_fmaf(d, d, a);
        } else {
            const float d = q - v;
            a = __builtin_fabsf(d) + a;                  // andnot(-0.0) then add         simple_avx.rs:98
        }
    }
    static __device__ __forceinline__ void row_aux(acc_t (&)[1], const uint4 &) {}
    static __device__ __forceinline__ void mac(acc_t (&a)[NACC], const uint4 &q, const uint4 &v) {
        mac1(a[0], __uint_as_float(q.x), __uint_as_float(v.x));
        mac1(a[1], __uint_as_float(q.y), __uint_as_float(v.y));
        mac1(a[2], __uint_as_float(q.z), __uint_as_float(v.z));
        mac1(a[3], __uint_as_float(q.w), __uint_as_float(v.w));
    }
    // four_way_hsum + hsum256_ps_avx (simple_avx.rs:10-28), then the scalar tail (:208-211)
    static __device__ __forceinline__ float finish(acc_t (&a)[NACC], acc_t (&)[1], const unsigned char *q_lds,
                                                   const unsigned char *row, uint32_t, const ScanArgs &args) {
        float lr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float s12 = a[k] + dpp_f32<DPP_QUAD_XOR1>(a[k]);        // sum1 = a+b | sum2 = c+d
            const float tot = s12 + dpp_f32<DPP_QUAD_XOR2>(s12);          // total = sum1 + sum2
            lr[k] = tot + dpp_f32<DPP_ROW_HALF_MIRROR>(tot);              // lr_sum = hi128 + lo128
        }
        float result = (lr[0] + lr[1]) + (lr[2] + lr[3]);                 // hadd, then p1 + p2
        if (args.tail_start < args.dim) {
            const float *qf = reinterpret_cast<const float *>(q_lds);
            const float *vf = reinterpret_cast<const float *>(row);
            for (uint32_t i = args.tail_start; i < args.dim; ++i) {
                if (METRIC == M_DOT) result += qf[i] * vf[i];
                else if (METRIC == M_EUCLID) { const float d = qf[i] - vf[i]; result += d * d; }
                else result += __builtin_fabsf(qf[i] - vf[i]);
            }
        }
        return METRIC == M_DOT ? result : -result;
    }
};

// ------------------------------------------------------------------------------------------
// f16 : each lane holds 8 halfs of the row and of the query per step; every element is widened
// to f32 (exact) and accumulated with f32 FMA like the reference (metric_f16/avx/dot.rs:29-52,
// euclid.rs, manhattan.rs).  The 32 AVX partial sums are NOT kept apart (that would need the two
// 64-byte halves of a cache line in different lanes): summation order differs from x86, scores
// agree within 1e-5 relative to sum(abs(terms)) (the reference's own SIMD-vs-scalar test allows 5e-4).
// ------------------------------------------------------------------------------------------
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

template <int METRIC>
struct RowF16 {
    static constexpr int NACC = 2;
    static constexpr int NRAUX = 0;
    static constexpr int R16 = 2;
    typedef float acc_t;

    static __device__ __forceinline__ void mac2(float &a, uint32_t q, uint32_t v) {
        const half2_t qh = *reinterpret_cast<const half2_t *>(&q);
        const half2_t vh = *reinterpret_cast<const half2_t *>(&v);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float x = (float)qh[i], y = (float)vh[i];
            if (METRIC == M_DOT) a = __builtin_fmaf(x, y, a);
            else if (METRIC == M_EUCLID) { const float d = x - y; a = __builtin_fmaf(d, d, a); }
            else a = __builtin_fabsf(x - y) + a;
        }
    }
    static __device__ __forceinline__ void row_aux(acc_t (&)[1], const uint4 &) {}
    static __device__ __forceinline__ void mac(acc_t (&a)[NACC], const uint4 &q, const uint4 &v) {
        mac2(a[0], q.x, v.x);
        mac2(a[1], q.y, v.y);
        mac2(a[0], q.z, v.z);
        mac2(a[1], q.w, v.w);
    }
    static __device__ __forceinline__ float finish(acc_t (&a)[NACC], acc_t (&)[1], const unsigned char *q_lds,
                                                   const unsigned char *row, uint32_t, const ScanArgs &args) {
        float result = reduce8_f32(a[0] + a[1]);
        if (args.tail_start < args.dim) {   // scalar tail, mul then add (avx/dot.rs:64-66)
            const _Float16 *qh = reinterpret_cast<const _Float16 *>(q_lds);
            const _Float16 *vh = reinterpret_cast<const _Float16 *>(row);
            for (uint32_t i = args.tail_start; i < args.dim; ++i) {
                const float x = (float)qh[i], y = (float)vh[i];
                if (METRIC == M_DOT) result += x * y;
                else if (METRIC == M_EUCLID) { const float d = x - y; result += d * d; }
                else result += __builtin_fabsf(x - y);
            }
        }
        return METRIC == M_DOT ? result : -result;
    }
};

// ------------------------------------------------------------------------------------------
// u8 : exact integer lanes.  A lane's 16-byte piece p covers bytes 16*(p%2) .. +15 of a 32-byte AVX2
// block, i.e. i32 lanes 4*(p%2)+k for dword k (avx2/dot.rs:36-52: madd_epi16 of even bytes + of odd
// bytes lands bytes 4j..4j+3 in i32 lane j).  piece = 2r + h  =>  a quad (same h) owns lanes 4h..4h+3.
// Final conversion: AVX2 order (8 x cvtepi32_ps, hsum256) or scalar order (one i32 -> f32 cast).
// ------------------------------------------------------------------------------------------
template <int METRIC>
struct RowU8 {
    static constexpr int NACC = METRIC == M_EUCLID ? 8 : 4;
    static constexpr int NRAUX = METRIC == M_COSINE ? 4 : 0;
    static constexpr int R16 = METRIC == M_EUCLID ? 1 : 2;
    typedef uint32_t acc_t;

    static __device__ __forceinline__ void row_aux(acc_t (&ra)[NRAUX > 0 ? NRAUX : 1], const uint4 &v) {
        if (METRIC == M_COSINE) {   // norm2 of the stored row (cosine.rs:47-58), once per row
            ra[0] = __builtin_amdgcn_udot4(v.x, v.x, ra[0], false);
            ra[1] = __builtin_amdgcn_udot4(v.y, v.y, ra[1], false);
            ra[2] = __builtin_amdgcn_udot4(v.z, v.z, ra[2], false);
            ra[3] = __builtin_amdgcn_udot4(v.w, v.w, ra[3], false);
        }
    }
    static __device__ __forceinline__ void mac(acc_t (&a)[NACC], const uint4 &q, const uint4 &v) {
        const uint32_t qq[4] = {q.x, q.y, q.z, q.w}, vv[4] = {v.x, v.y, v.z, v.w};
        if (METRIC == M_DOT || METRIC == M_COSINE) {
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = __builtin_amdgcn_udot4(qq[k], vv[k], a[k], false);
        } else if (METRIC == M_EUCLID) {
            // sum (q-v)^2 = sum q^2 + sum v^2 - 2 sum qv, all exact in u32 (euclid.rs:33-46)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a[k] = __builtin_amdgcn_udot4(qq[k], qq[k], a[k], false);
                a[k] = __builtin_amdgcn_udot4(vv[k], vv[k], a[k], false);
                a[4 + k] = __builtin_amdgcn_udot4(qq[k], vv[k], a[4 + k], false);
            }
        } else {
            // _mm256_sad_epu8: 8-byte groups land in the even i32 lanes (manhattan.rs:33-35)
            a[0] = __builtin_amdgcn_sad_u8(qq[0], vv[0], a[0]);
            a[0] = __builtin_amdgcn_sad_u8(qq[1], vv[1], a[0]);
            a[2] = __builtin_amdgcn_sad_u8(qq[2], vv[2], a[2]);
            a[2] = __builtin_amdgcn_sad_u8(qq[3], vv[3], a[2]);
        }
    }
    // 8 exact i32 lanes -> f32 in hsum256_ps_avx order (simple_avx.rs:10-16)
    static __device__ __forceinline__ float avx_hsum_i32(const uint32_t (&lane_tot)[4]) {
        float lr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float f = (float)(int32_t)lane_tot[k];               // _mm256_cvtepi32_ps
            lr[k] = f + dpp_f32<DPP_ROW_HALF_MIRROR>(f);               // hi128 + lo128
        }
        return (lr[0] + lr[1]) + (lr[2] + lr[3]);
    }
    static __device__ __forceinline__ float finish(acc_t (&a)[NACC], acc_t (&ra)[NRAUX > 0 ? NRAUX : 1],
                                                   const unsigned char *q_lds, const unsigned char *row, uint32_t,
                                                   const ScanArgs &args) {
        const QueryAux *aux = reinterpret_cast<const QueryAux *>(q_lds + args.aux_off);
        return finish_with(a, ra, q_lds, row, args, METRIC == M_COSINE ? aux->f0 : 0.0f, METRIC == M_COSINE ? aux->i0 : 0);
    }
    // norm1_f / norm1_i: the query's sum of squares (AVX2 order as f32 / scalar order as i32), only read by the cosine
    static __device__ __forceinline__ float finish_with(acc_t (&a)[NACC], acc_t (&ra)[NRAUX > 0 ? NRAUX : 1],
                                                        const unsigned char *q_lds, const unsigned char *row,
                                                        const ScanArgs &args, float norm1_f, int32_t norm1_i) {
        uint32_t lt[4], nt[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t x = a[k];
            if (METRIC == M_EUCLID) x = a[k] - 2u * a[4 + k];
            lt[k] = reduce4_u32(x);
            if (METRIC == M_COSINE) nt[k] = reduce4_u32(ra[k]);
        }
        // the reference's remainder loop (len % 32 bytes), exact i32
        int32_t rem = 0, rem_n2 = 0;
        const bool has_rem = args.tail_start < args.dim;
        for (uint32_t i = args.tail_start; i < args.dim; ++i) {
            const int32_t x = q_lds[i], y = row[i];
            if (METRIC == M_DOT || METRIC == M_COSINE) rem += x * y;
            else if (METRIC == M_EUCLID) rem += (x - y) * (x - y);
            else rem += x > y ? x - y : y - x;
            if (METRIC == M_COSINE) rem_n2 += y * y;
        }
        if (args.flags & QMX_SEG_U8_SCALAR_ORDER) {
            // metric_uint/simple_*.rs: one i32 total, cast once
            const int32_t tot = (int32_t)((lt[0] + lt[1]) + (lt[2] + lt[3]));
            const int32_t all = tot + dpp_i32<DPP_ROW_HALF_MIRROR>(tot) + rem;
            if (METRIC == M_COSINE) {
                const int32_t n2h = (int32_t)((nt[0] + nt[1]) + (nt[2] + nt[3]));
                const int32_t n2 = n2h + dpp_i32<DPP_ROW_HALF_MIRROR>(n2h) + rem_n2;
                const int32_t n1 = norm1_i;
                if (n1 == 0 || n2 == 0) return 0.0f;                       // simple_cosine.rs:71-73
                return (float)all / __builtin_sqrtf((float)n1 * (float)n2);
            }
            return (METRIC == M_DOT) ? (float)all : -(float)all;
        }
        float score = avx_hsum_i32(lt);
        if (has_rem) score += (float)rem;
        if (METRIC == M_COSINE) {
            float norm2 = avx_hsum_i32(nt);
            if (has_rem) norm2 += (float)rem_n2;
            const float denominator = norm1_f * norm2;                     // norm1 * norm2, cosine.rs:103-108
            if (denominator == 0.0f) return 0.0f;
            return score / __builtin_sqrtf(denominator);
        }
        return (METRIC == M_DOT) ? score : -score;
    }
};

// The same row policy with a STORED ROW as the query (HNSW build over a u8 segment, FilteredScorer::new_internal): the per-pair
// cosine (metric_uint/avx2/cosine.rs:47-108, simple_cosine.rs) needs the query's norm, which the aux block of a query entry carries
// and a bare stored row does not: it arrives in ScanArgs::u8_qnorm_* from the per-row norm column (hnsw_build.hpp query_args).
template <int METRIC>
struct RowU8Internal : RowU8<METRIC> {
    typedef RowU8<METRIC> B;
    static constexpr bool INTERNAL_NORM = METRIC == M_COSINE;
    static __device__ __forceinline__ float finish(typename B::acc_t (&a)[B::NACC], typename B::acc_t (&ra)[B::NRAUX > 0 ? B::NRAUX : 1],
                                                   const unsigned char *q_lds, const unsigned char *row, uint32_t, const ScanArgs &args) {
        return B::finish_with(a, ra, q_lds, row, args, args.u8_qnorm_f, args.u8_qnorm_i);
    }
};

// ------------------------------------------------------------------------------------------
// f32 below the AVX threshold: SSE leaf for 16 <= dim < 32 (spaces/simple_sse.rs:19-243: 4 x __m128,
// mul THEN add, hsum128 of each register, scalar adds, scalar tail), scalar leaf below 16
// (spaces/simple.rs:214-239: sequential sum from -0.0).
// ------------------------------------------------------------------------------------------
template <int METRIC>
__device__ __forceinline__ float term_f32(float q, float v) {
    if (METRIC == M_DOT) return q * v;
    const float d = q - v;
    return METRIC == M_EUCLID ? d * d : __builtin_fabsf(d);
}
// the SSE / scalar leaf over f32 values produced by `load(i)`
template <int METRIC, class LoadQ, class LoadV>
__device__ __forceinline__ float small_f32_leaf(uint32_t dim, LoadQ lq, LoadV lv) {
    float result;
    uint32_t i0 = 0;
    if (dim >= 16) {
        float h[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = term_f32<METRIC>(lq(4 * r + k), lv(4 * r + k)) + 0.0f;  // add_ps(term, zero)
            h[r] = (x[0] + x[2]) + (x[1] + x[3]);                                                  // hsum128_ps_sse
        }
        result = ((h[0] + h[1]) + h[2]) + h[3];
        i0 = 16;
    } else {
        result = -0.0f;
    }
    for (uint32_t i = i0; i < dim; ++i) result += term_f32<METRIC>(lq(i), lv(i));
    return METRIC == M_DOT ? result : -result;
}

template <int METRIC>
struct SmallF32 {
    static __device__ float score(const unsigned char *qb, const unsigned char *rb, uint32_t, const ScanArgs &a) {
        const float *q = reinterpret_cast<const float *>(qb);
        const float *v = reinterpret_cast<const float *>(rb);
        return small_f32_leaf<METRIC>(a.dim, [&](uint32_t i) { return q[i]; }, [&](uint32_t i) { return v[i]; });
    }
};
// f16 below 32: the SSE leaf widens both vectors to f32 and calls the f32 SSE kernel
// (metric_f16/sse/dot.rs:10-17), the scalar leaf multiplies widened values (simple_dot.rs:59-67)
template <int METRIC>
struct SmallF16 {
    static __device__ float score(const unsigned char *qb, const unsigned char *rb, uint32_t, const ScanArgs &a) {
        const _Float16 *q = reinterpret_cast<const _Float16 *>(qb);
        const _Float16 *v = reinterpret_cast<const _Float16 *>(rb);
        return small_f32_leaf<METRIC>(a.dim, [&](uint32_t i) { return (float)q[i]; }, [&](uint32_t i) { return (float)v[i]; });
    }
};
// u8 below 32: SSE2 leaf (metric_uint/sse2/*.rs: one 16-byte step, 4 i32 lanes = the 4 dwords,
// cvtepi32_ps, hsum128, remainder added as f32) or the scalar leaf (simple_*.rs)
template <int METRIC>
struct SmallU8 {
    static __device__ float score(const unsigned char *q, const unsigned char *v, uint32_t, const ScanArgs &a) {
        const uint32_t dim = a.dim;
        const bool sse = dim >= 16 && !(a.flags & QMX_SEG_U8_SCALAR_ORDER);
        int32_t lane[4] = {0, 0, 0, 0}, n1l[4] = {0, 0, 0, 0}, n2l[4] = {0, 0, 0, 0};
        int32_t rem = 0, r1 = 0, r2 = 0;
        const uint32_t body = sse ? 16 : 0;
        for (uint32_t i = 0; i < dim; ++i) {
            const int32_t x = q[i], y = v[i];
            int32_t t;
            if (METRIC == M_DOT || METRIC == M_COSINE) t = x * y;
            else if (METRIC == M_EUCLID) t = (x - y) * (x - y);
            else t = x > y ? x - y : y - x;
            if (i < body) {
                // sad_epu8 puts each 8-byte group into an even lane (lanes 0 and 2)
                const int l = METRIC == M_MANHATTAN ? (int)(i >> 3) * 2 : (int)(i >> 2);
                lane[l] += t; n1l[l] += x * x; n2l[l] += y * y;
            } else {
                rem += t; r1 += x * x; r2 += y * y;
            }
        }
        if (!sse) {
            if (METRIC == M_COSINE) {
                if (r1 == 0 || r2 == 0) return 0.0f;
                return (float)rem / __builtin_sqrtf((float)r1 * (float)r2);
            }
            return METRIC == M_DOT ? (float)rem : -(float)rem;
        }
        auto hsum = [](const int32_t(&l)[4]) { return ((float)l[0] + (float)l[2]) + ((float)l[1] + (float)l[3]); };
        const bool has_rem = dim > 16;
        float s = hsum(lane);
        if (has_rem) s += (float)rem;
        if (METRIC == M_COSINE) {
            float n1 = hsum(n1l), n2 = hsum(n2l);
            if (has_rem) { n1 += (float)r1; n2 += (float)r2; }
            const float den = n1 * n2;
            if (den == 0.0f) return 0.0f;
            return s / __builtin_sqrtf(den);
        }
        return METRIC == M_DOT ? s : -s;
    }
};

// ------------------------------------------------------------------------------------------
// dispatch
// ------------------------------------------------------------------------------------------
template <template <int> class Row, template <int> class Small, bool COSINE_IS_DOT, class L>
static int32_t dispatch_metric(const L &l, int distance, const ScanArgs &a) {
    const bool small = a.dim < 32;
    switch (distance) {
        case QMX_DISTANCE_COSINE:
            if constexpr (!COSINE_IS_DOT)
                return small ? l.template small<Small<M_COSINE>>(a) : l.template row<Row<M_COSINE>>(a);
            // CosineMetric::similarity == DotProductMetric::similarity on normalised vectors (simple.rs:174-176)
        case QMX_DISTANCE_DOT: return small ? l.template small<Small<M_DOT>>(a) : l.template row<Row<M_DOT>>(a);
        case QMX_DISTANCE_EUCLID: return small ? l.template small<Small<M_EUCLID>>(a) : l.template row<Row<M_EUCLID>>(a);
        case QMX_DISTANCE_MANHATTAN: return small ? l.template small<Small<M_MANHATTAN>>(a) : l.template row<Row<M_MANHATTAN>>(a);
    }
    set_error("bad distance %d", distance);
    return QMX_ERR_BAD_ARG;
}

template <class L>
static int32_t dispatch_dense(const L &l, int dtype, int distance, const ScanArgs &a) {
    switch (dtype) {
        case QMX_DTYPE_F32: return dispatch_metric<RowF32, SmallF32, true>(l, distance, a);
        case QMX_DTYPE_F16: return dispatch_metric<RowF16, SmallF16, true>(l, distance, a);
        case QMX_DTYPE_U8: return dispatch_metric<RowU8, SmallU8, false>(l, distance, a);
    }
    set_error("scan: dtype %d / distance %d not supported", dtype, distance);
    return QMX_ERR_NOT_SUPPORTED;
}

}  // namespace qmx

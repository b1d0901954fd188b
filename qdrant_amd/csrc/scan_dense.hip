// scan_dense.hip — Metric<f32|f16|u8>::similarity over a device-resident dense block.
//
// Reference leaves restated as lane policies (one policy = one "SIMD leaf" of SURVEY §2.2):
//   f32  lib/segment/src/spaces/simple_avx.rs:32-213        (dot / euclid / manhattan, AVX+FMA order)
//   f16  lib/segment/src/spaces/metric_f16/avx/*.rs          (F16C convert, f32 FMA, 4 hsums then adds)
//   u8   lib/segment/src/spaces/metric_uint/avx2/*.rs        (exact i32 lanes, cvtepi32_ps, f32 hsum)
//        lib/segment/src/spaces/metric_uint/simple_*.rs      (scalar order, QMX_SEG_U8_SCALAR_ORDER)
// Compiled with -ffp-contract=off: every fused multiply-add below is an explicit fmaf, every
// separate mul/add stays separate, as in the Rust/C reference.
#include "scan_common.hpp"

namespace qmx {

enum { M_DOT = 0, M_EUCLID = 1, M_MANHATTAN = 2, M_COSINE = 3 };

// ------------------------------------------------------------------------------------------
// f32 : 8 lanes x float4 = the 4 x __m256 accumulators of one 32-float AVX iteration.
// lane (h = t>>2, r = t&3) holds AVX register r, SIMD lanes 4h..4h+3.
// ------------------------------------------------------------------------------------------
template <int METRIC>
struct RowF32 {
    static constexpr int ELEM = 4;
    static constexpr int SEG = 128;
    static constexpr int NACC = 4;
    typedef float acc_t;
    typedef float4 vec_t;

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
    static __device__ __forceinline__ void mac(acc_t (&a)[NACC], const vec_t &q, const vec_t &v) {
        mac1(a[0], q.x, v.x);
        mac1(a[1], q.y, v.y);
        mac1(a[2], q.z, v.z);
        mac1(a[3], q.w, v.w);
    }
    // four_way_hsum + hsum256_ps_avx (simple_avx.rs:10-28), then the scalar tail (:208-211)
    static __device__ __forceinline__ float finish(acc_t (&a)[NACC], const unsigned char *q_lds,
                                                   const unsigned char *row, uint32_t, uint32_t nseg,
                                                   const ScanArgs &args) {
        float lr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float s12 = a[k] + dpp_f32<DPP_QUAD_XOR1>(a[k]);        // sum1 = a+b | sum2 = c+d
            const float tot = s12 + dpp_f32<DPP_QUAD_XOR2>(s12);          // total = sum1 + sum2
            lr[k] = tot + dpp_f32<DPP_ROW_HALF_MIRROR>(tot);              // lr_sum = hi128 + lo128
        }
        float result = (lr[0] + lr[1]) + (lr[2] + lr[3]);                 // hadd, then p1 + p2
        const uint32_t m = nseg * 32;
        if (m < args.dim) {
            const float *qf = reinterpret_cast<const float *>(q_lds);
            const float *vf = reinterpret_cast<const float *>(row);
            for (uint32_t i = m; i < args.dim; ++i) {
                if (METRIC == M_DOT) result += qf[i] * vf[i];
                else if (METRIC == M_EUCLID) { const float d = qf[i] - vf[i]; result += d * d; }
                else result += __builtin_fabsf(qf[i] - vf[i]);
            }
        }
        return METRIC == M_DOT ? result : -result;
    }
};


// ------------------------------------------------------------------------------------------
// f32 below the AVX threshold: SSE leaf for 16 <= dim < 32 (spaces/simple_sse.rs:19-243: 4 x __m128,
// mul THEN add, hsum128 of each register, scalar adds, scalar tail), scalar leaf below 16
// (spaces/simple.rs:214-239: sequential sum from -0.0).
// ------------------------------------------------------------------------------------------
template <int METRIC>
struct SmallF32 {
    static __device__ __forceinline__ float term(float q, float v) {
        if (METRIC == M_DOT) return q * v;
        const float d = q - v;
        return METRIC == M_EUCLID ? d * d : __builtin_fabsf(d);
    }
    static __device__ float score(const unsigned char *qb, const unsigned char *rb, uint32_t, const ScanArgs &a) {
        const float *q = reinterpret_cast<const float *>(qb);
        const float *v = reinterpret_cast<const float *>(rb);
        const uint32_t dim = a.dim;
        float result;
        uint32_t i0 = 0;
        if (dim >= 16) {
            float h[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) x[k] = term(q[4 * r + k], v[4 * r + k]) + 0.0f;  // _mm_add_ps(term, zero)
                h[r] = (x[0] + x[2]) + (x[1] + x[3]);                                        // hsum128_ps_sse
            }
            result = ((h[0] + h[1]) + h[2]) + h[3];
            i0 = 16;
        } else {
            result = -0.0f;
        }
        for (uint32_t i = i0; i < dim; ++i) result += term(q[i], v[i]);
        return METRIC == M_DOT ? result : -result;
    }
};

// ------------------------------------------------------------------------------------------
// dispatch
// ------------------------------------------------------------------------------------------
template <class P, int QT, int R, int U>
static int32_t launch_qt(hipStream_t st, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid) {
    const bool ids = a.ids != nullptr;
    if (mode == SCAN_TOPK) {
        return ids ? launch_scan_inst<P, QT, R, U, true, SCAN_TOPK>(st, a, num_cus, grid)
                   : launch_scan_inst<P, QT, R, U, false, SCAN_TOPK>(st, a, num_cus, grid);
    }
    return ids ? launch_scan_inst<P, QT, R, U, true, SCAN_SCORES>(st, a, num_cus, grid)
               : launch_scan_inst<P, QT, R, U, false, SCAN_SCORES>(st, a, num_cus, grid);
}

template <class P>
static int32_t launch_policy(hipStream_t st, int qt, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid) {
    switch (qt) {
        case 1: return launch_qt<P, 1, 4, 4>(st, mode, a, num_cus, grid);
        case 2: return launch_qt<P, 2, 4, 2>(st, mode, a, num_cus, grid);
        case 4: return launch_qt<P, 4, 2, 4>(st, mode, a, num_cus, grid);
        case 8: return launch_qt<P, 8, 2, 2>(st, mode, a, num_cus, grid);
        case 16: return launch_qt<P, 16, 2, 2>(st, mode, a, num_cus, grid);
        default: set_error("unsupported query tile %d", qt); return QMX_ERR_BAD_ARG;
    }
}

int32_t launch_scan_dense(hipStream_t st, int dtype, int distance, int qt, ScanMode mode,
                          const ScanArgs &a, int num_cus, uint32_t *grid_out) {
    if (dtype == QMX_DTYPE_F32 && a.dim < 32) {
        switch (distance) {
            case QMX_DISTANCE_COSINE:
            case QMX_DISTANCE_DOT: return launch_small<SmallF32<M_DOT>>(st, mode, a, num_cus, grid_out);
            case QMX_DISTANCE_EUCLID: return launch_small<SmallF32<M_EUCLID>>(st, mode, a, num_cus, grid_out);
            case QMX_DISTANCE_MANHATTAN: return launch_small<SmallF32<M_MANHATTAN>>(st, mode, a, num_cus, grid_out);
        }
    }
    if (dtype == QMX_DTYPE_F32) {
        switch (distance) {
            case QMX_DISTANCE_COSINE:  // CosineMetric::similarity == DotProductMetric::similarity (simple.rs:174-176)
            case QMX_DISTANCE_DOT: return launch_policy<RowF32<M_DOT>>(st, qt, mode, a, num_cus, grid_out);
            case QMX_DISTANCE_EUCLID: return launch_policy<RowF32<M_EUCLID>>(st, qt, mode, a, num_cus, grid_out);
            case QMX_DISTANCE_MANHATTAN: return launch_policy<RowF32<M_MANHATTAN>>(st, qt, mode, a, num_cus, grid_out);
        }
    }
    set_error("scan: dtype %d / distance %d not supported", dtype, distance);
    return QMX_ERR_NOT_SUPPORTED;
}

}  // namespace qmx

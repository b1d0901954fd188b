/*
 * qdrant_oracle.c — CPU restatement of qdrant v1.19.0's vector-scoring hot path.
 * TEST INFRASTRUCTURE ONLY (see qdrant_oracle.h).  Build: oracle/Makefile
 *   gcc -O2 -march=haswell -mf16c -ffp-contract=off -fPIC -shared
 * Paths in comments are relative to the qdrant source tree.
 */
#include "qdrant_oracle.h"

#include <immintrin.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * f32 metrics
 * ---------------------------------------------------------------------------------------- */

/* lib/segment/src/spaces/simple_avx.rs:10-16 */
static inline float hsum256_ps_avx(__m256 x) {
    __m128 lr_sum = _mm_add_ps(_mm256_extractf128_ps(x, 1), _mm256_castps256_ps128(x));
    __m128 hsum = _mm_hadd_ps(lr_sum, lr_sum);
    float p1 = _mm_cvtss_f32(hsum);
    float p2 = _mm_cvtss_f32(_mm_shuffle_ps(hsum, hsum, 0x55));
    return p1 + p2;
}
/* simple_avx.rs:21-28 */
static inline float four_way_hsum(__m256 a, __m256 b, __m256 c, __m256 d) {
    __m256 sum1 = _mm256_add_ps(a, b);
    __m256 sum2 = _mm256_add_ps(c, d);
    __m256 total = _mm256_add_ps(sum1, sum2);
    return hsum256_ps_avx(total);
}
/* simple_sse.rs:13-17 */
static inline float hsum128_ps_sse(__m128 x) {
    __m128 x64 = _mm_add_ps(x, _mm_movehl_ps(x, x));
    __m128 x32 = _mm_add_ss(x64, _mm_shuffle_ps(x64, x64, 0x55));
    return _mm_cvtss_f32(x32);
}

/* Rust `iter.sum::<f32>()` folds from -0.0 (core::iter::Sum for f32, rustc >= 1.83); only the
 * sign of an all-(-0.0) sum depends on it. */
#define RUST_SUM_INIT (-0.0f)

/* simple_avx.rs:169-213 */
static float dot_similarity_avx(const float *v1, const float *v2, size_t n) {
    size_t m = n - (n % 32);
    const float *p1 = v1, *p2 = v2;
    __m256 s1 = _mm256_setzero_ps(), s2 = s1, s3 = s1, s4 = s1;
    for (size_t i = 0; i < m; i += 32) {
        s1 = _mm256_fmadd_ps(_mm256_loadu_ps(p1), _mm256_loadu_ps(p2), s1);
        s2 = _mm256_fmadd_ps(_mm256_loadu_ps(p1 + 8), _mm256_loadu_ps(p2 + 8), s2);
        s3 = _mm256_fmadd_ps(_mm256_loadu_ps(p1 + 16), _mm256_loadu_ps(p2 + 16), s3);
        s4 = _mm256_fmadd_ps(_mm256_loadu_ps(p1 + 24), _mm256_loadu_ps(p2 + 24), s4);
        p1 += 32; p2 += 32;
    }
    float result = four_way_hsum(s1, s2, s3, s4);
    for (size_t i = 0; i < n - m; i++) result += p1[i] * p2[i];
    return result;
}
/* simple_avx.rs:32-75 */
static float euclid_similarity_avx(const float *v1, const float *v2, size_t n) {
    size_t m = n - (n % 32);
    const float *p1 = v1, *p2 = v2;
    __m256 s1 = _mm256_setzero_ps(), s2 = s1, s3 = s1, s4 = s1;
    for (size_t i = 0; i < m; i += 32) {
        __m256 d1 = _mm256_sub_ps(_mm256_loadu_ps(p1), _mm256_loadu_ps(p2));
        s1 = _mm256_fmadd_ps(d1, d1, s1);
        __m256 d2 = _mm256_sub_ps(_mm256_loadu_ps(p1 + 8), _mm256_loadu_ps(p2 + 8));
        s2 = _mm256_fmadd_ps(d2, d2, s2);
        __m256 d3 = _mm256_sub_ps(_mm256_loadu_ps(p1 + 16), _mm256_loadu_ps(p2 + 16));
        s3 = _mm256_fmadd_ps(d3, d3, s3);
        __m256 d4 = _mm256_sub_ps(_mm256_loadu_ps(p1 + 24), _mm256_loadu_ps(p2 + 24));
        s4 = _mm256_fmadd_ps(d4, d4, s4);
        p1 += 32; p2 += 32;
    }
    float result = four_way_hsum(s1, s2, s3, s4);
    for (size_t i = 0; i < n - m; i++) { float d = p1[i] - p2[i]; result += d * d; }
    return -result;
}
/* simple_avx.rs:79-123 */
static float manhattan_similarity_avx(const float *v1, const float *v2, size_t n) {
    const __m256 mask = _mm256_set1_ps(-0.0f);
    size_t m = n - (n % 32);
    const float *p1 = v1, *p2 = v2;
    __m256 s1 = _mm256_setzero_ps(), s2 = s1, s3 = s1, s4 = s1;
    for (size_t i = 0; i < m; i += 32) {
        __m256 d1 = _mm256_sub_ps(_mm256_loadu_ps(p1), _mm256_loadu_ps(p2));
        s1 = _mm256_add_ps(_mm256_andnot_ps(mask, d1), s1);
        __m256 d2 = _mm256_sub_ps(_mm256_loadu_ps(p1 + 8), _mm256_loadu_ps(p2 + 8));
        s2 = _mm256_add_ps(_mm256_andnot_ps(mask, d2), s2);
        __m256 d3 = _mm256_sub_ps(_mm256_loadu_ps(p1 + 16), _mm256_loadu_ps(p2 + 16));
        s3 = _mm256_add_ps(_mm256_andnot_ps(mask, d3), s3);
        __m256 d4 = _mm256_sub_ps(_mm256_loadu_ps(p1 + 24), _mm256_loadu_ps(p2 + 24));
        s4 = _mm256_add_ps(_mm256_andnot_ps(mask, d4), s4);
        p1 += 32; p2 += 32;
    }
    float result = four_way_hsum(s1, s2, s3, s4);
    for (size_t i = 0; i < n - m; i++) result += fabsf(p1[i] - p2[i]);
    return -result;
}
/* spaces/tools.rs:14-16 */
static inline int is_length_zero_or_normalized(float length) {
    return length < 1.1920929e-7f /* f32::EPSILON */ || fabsf(length - 1.0f) <= 1.0e-6f;
}
/* simple_avx.rs:127-165 */
static void cosine_preprocess_avx(const float *v, float *out, size_t n) {
    size_t m = n - (n % 32);
    const float *p = v;
    __m256 s1 = _mm256_setzero_ps(), s2 = s1, s3 = s1, s4 = s1;
    for (size_t i = 0; i < m; i += 32) {
        __m256 a = _mm256_loadu_ps(p);      s1 = _mm256_fmadd_ps(a, a, s1);
        __m256 b = _mm256_loadu_ps(p + 8);  s2 = _mm256_fmadd_ps(b, b, s2);
        __m256 c = _mm256_loadu_ps(p + 16); s3 = _mm256_fmadd_ps(c, c, s3);
        __m256 d = _mm256_loadu_ps(p + 24); s4 = _mm256_fmadd_ps(d, d, s4);
        p += 32;
    }
    float length = four_way_hsum(s1, s2, s3, s4);
    for (size_t i = 0; i < n - m; i++) length += p[i] * p[i];
    if (is_length_zero_or_normalized(length)) { if (out != v) memmove(out, v, n * sizeof(float)); return; }
    length = sqrtf(length);
    for (size_t i = 0; i < n; i++) out[i] = v[i] / length;
}

/* simple_sse.rs:197-243 */
static float dot_similarity_sse(const float *v1, const float *v2, size_t n) {
    size_t m = n - (n % 16);
    const float *p1 = v1, *p2 = v2;
    __m128 s1 = _mm_setzero_ps(), s2 = s1, s3 = s1, s4 = s1;
    for (size_t i = 0; i < m; i += 16) {
        s1 = _mm_add_ps(_mm_mul_ps(_mm_loadu_ps(p1), _mm_loadu_ps(p2)), s1);
        s2 = _mm_add_ps(_mm_mul_ps(_mm_loadu_ps(p1 + 4), _mm_loadu_ps(p2 + 4)), s2);
        s3 = _mm_add_ps(_mm_mul_ps(_mm_loadu_ps(p1 + 8), _mm_loadu_ps(p2 + 8)), s3);
        s4 = _mm_add_ps(_mm_mul_ps(_mm_loadu_ps(p1 + 12), _mm_loadu_ps(p2 + 12)), s4);
        p1 += 16; p2 += 16;
    }
    float result = hsum128_ps_sse(s1) + hsum128_ps_sse(s2) + hsum128_ps_sse(s3) + hsum128_ps_sse(s4);
    for (size_t i = 0; i < n - m; i++) result += p1[i] * p2[i];
    return result;
}
/* simple_sse.rs:19-59 */
static float euclid_similarity_sse(const float *v1, const float *v2, size_t n) {
    size_t m = n - (n % 16);
    const float *p1 = v1, *p2 = v2;
    __m128 s1 = _mm_setzero_ps(), s2 = s1, s3 = s1, s4 = s1;
    for (size_t i = 0; i < m; i += 16) {
        __m128 d1 = _mm_sub_ps(_mm_loadu_ps(p1), _mm_loadu_ps(p2));
        s1 = _mm_add_ps(_mm_mul_ps(d1, d1), s1);
        __m128 d2 = _mm_sub_ps(_mm_loadu_ps(p1 + 4), _mm_loadu_ps(p2 + 4));
        s2 = _mm_add_ps(_mm_mul_ps(d2, d2), s2);
        __m128 d3 = _mm_sub_ps(_mm_loadu_ps(p1 + 8), _mm_loadu_ps(p2 + 8));
        s3 = _mm_add_ps(_mm_mul_ps(d3, d3), s3);
        __m128 d4 = _mm_sub_ps(_mm_loadu_ps(p1 + 12), _mm_loadu_ps(p2 + 12));
        s4 = _mm_add_ps(_mm_mul_ps(d4, d4), s4);
        p1 += 16; p2 += 16;
    }
    float result = hsum128_ps_sse(s1) + hsum128_ps_sse(s2) + hsum128_ps_sse(s3) + hsum128_ps_sse(s4);
    for (size_t i = 0; i < n - m; i++) { float d = p1[i] - p2[i]; result += d * d; }
    return -result;
}
/* simple_sse.rs:61-105 */
static float manhattan_similarity_sse(const float *v1, const float *v2, size_t n) {
    const __m128 mask = _mm_set1_ps(-0.0f);
    size_t m = n - (n % 16);
    const float *p1 = v1, *p2 = v2;
    __m128 s1 = _mm_setzero_ps(), s2 = s1, s3 = s1, s4 = s1;
    for (size_t i = 0; i < m; i += 16) {
        __m128 d1 = _mm_sub_ps(_mm_loadu_ps(p1), _mm_loadu_ps(p2));
        s1 = _mm_add_ps(_mm_andnot_ps(mask, d1), s1);
        __m128 d2 = _mm_sub_ps(_mm_loadu_ps(p1 + 4), _mm_loadu_ps(p2 + 4));
        s2 = _mm_add_ps(_mm_andnot_ps(mask, d2), s2);
        __m128 d3 = _mm_sub_ps(_mm_loadu_ps(p1 + 8), _mm_loadu_ps(p2 + 8));
        s3 = _mm_add_ps(_mm_andnot_ps(mask, d3), s3);
        __m128 d4 = _mm_sub_ps(_mm_loadu_ps(p1 + 12), _mm_loadu_ps(p2 + 12));
        s4 = _mm_add_ps(_mm_andnot_ps(mask, d4), s4);
        p1 += 16; p2 += 16;
    }
    float result = hsum128_ps_sse(s1) + hsum128_ps_sse(s2) + hsum128_ps_sse(s3) + hsum128_ps_sse(s4);
    for (size_t i = 0; i < n - m; i++) result += fabsf(p1[i] - p2[i]);
    return -result;
}
/* simple_sse.rs:107-150 (same shape as the AVX one with 4 x __m128, mul+add) */
static void cosine_preprocess_sse(const float *v, float *out, size_t n) {
    size_t m = n - (n % 16);
    const float *p = v;
    __m128 s1 = _mm_setzero_ps(), s2 = s1, s3 = s1, s4 = s1;
    for (size_t i = 0; i < m; i += 16) {
        __m128 a = _mm_loadu_ps(p);      s1 = _mm_add_ps(_mm_mul_ps(a, a), s1);
        __m128 b = _mm_loadu_ps(p + 4);  s2 = _mm_add_ps(_mm_mul_ps(b, b), s2);
        __m128 c = _mm_loadu_ps(p + 8);  s3 = _mm_add_ps(_mm_mul_ps(c, c), s3);
        __m128 d = _mm_loadu_ps(p + 12); s4 = _mm_add_ps(_mm_mul_ps(d, d), s4);
        p += 16;
    }
    float length = hsum128_ps_sse(s1) + hsum128_ps_sse(s2) + hsum128_ps_sse(s3) + hsum128_ps_sse(s4);
    for (size_t i = 0; i < n - m; i++) length += p[i] * p[i];
    if (is_length_zero_or_normalized(length)) { if (out != v) memmove(out, v, n * sizeof(float)); return; }
    length = sqrtf(length);
    for (size_t i = 0; i < n; i++) out[i] = v[i] / length;
}

/* simple.rs:214-239 (scalar) */
static float dot_similarity_scalar(const float *a, const float *b, size_t n) {
    float s = RUST_SUM_INIT;
    for (size_t i = 0; i < n; i++) s += a[i] * b[i];
    return s;
}
static float euclid_similarity_scalar(const float *a, const float *b, size_t n) {
    float s = RUST_SUM_INIT;
    for (size_t i = 0; i < n; i++) { float d = a[i] - b[i]; s += d * d; }
    return -s;
}
static float manhattan_similarity_scalar(const float *a, const float *b, size_t n) {
    float s = RUST_SUM_INIT;
    for (size_t i = 0; i < n; i++) s += fabsf(a[i] - b[i]);
    return -s;
}
static void cosine_preprocess_scalar(const float *v, float *out, size_t n) {
    float length = RUST_SUM_INIT;
    for (size_t i = 0; i < n; i++) length += v[i] * v[i];
    if (is_length_zero_or_normalized(length)) { if (out != v) memmove(out, v, n * sizeof(float)); return; }
    length = sqrtf(length);
    for (size_t i = 0; i < n; i++) out[i] = v[i] / length;
}

/* dispatch thresholds: simple.rs:15 (MIN_DIM_SIZE_AVX = 32), :22 (MIN_DIM_SIZE_SIMD = 16) */
static int pick_isa(int isa, size_t n) {
    if (isa != QO_ISA_AUTO) return isa;
    if (n >= 32) return QO_ISA_AVX;
    if (n >= 16) return QO_ISA_SSE;
    return QO_ISA_SCALAR;
}

float qo_dot_f32(const float *a, const float *b, size_t n, int isa) {
    switch (pick_isa(isa, n)) {
        case QO_ISA_AVX: return dot_similarity_avx(a, b, n);
        case QO_ISA_SSE: return dot_similarity_sse(a, b, n);
        default: return dot_similarity_scalar(a, b, n);
    }
}
float qo_euclid_f32(const float *a, const float *b, size_t n, int isa) {
    switch (pick_isa(isa, n)) {
        case QO_ISA_AVX: return euclid_similarity_avx(a, b, n);
        case QO_ISA_SSE: return euclid_similarity_sse(a, b, n);
        default: return euclid_similarity_scalar(a, b, n);
    }
}
float qo_manhattan_f32(const float *a, const float *b, size_t n, int isa) {
    switch (pick_isa(isa, n)) {
        case QO_ISA_AVX: return manhattan_similarity_avx(a, b, n);
        case QO_ISA_SSE: return manhattan_similarity_sse(a, b, n);
        default: return manhattan_similarity_scalar(a, b, n);
    }
}
float qo_similarity_f32(int distance, const float *q, const float *v, size_t n) {
    switch (distance) {
        case QO_COSINE: /* simple.rs:174-176: cosine similarity == dot on preprocessed vectors */
        case QO_DOT: return qo_dot_f32(q, v, n, QO_ISA_AUTO);
        case QO_EUCLID: return qo_euclid_f32(q, v, n, QO_ISA_AUTO);
        default: return qo_manhattan_f32(q, v, n, QO_ISA_AUTO);
    }
}
void qo_cosine_preprocess_f32(const float *in, float *out, size_t n, int isa) {
    switch (pick_isa(isa, n)) {
        case QO_ISA_AVX: cosine_preprocess_avx(in, out, n); break;
        case QO_ISA_SSE: cosine_preprocess_sse(in, out, n); break;
        default: cosine_preprocess_scalar(in, out, n); break;
    }
}
void qo_preprocess_f32(int distance, const float *in, float *out, size_t n) {
    if (distance == QO_COSINE) qo_cosine_preprocess_f32(in, out, n, QO_ISA_AUTO);
    else if (out != in) memmove(out, in, n * sizeof(float));
}
float qo_postprocess(int distance, float score) {
    switch (distance) {
        case QO_EUCLID: return sqrtf(fabsf(score));
        case QO_MANHATTAN: return fabsf(score);
        default: return score;
    }
}

/* ------------------------------------------------------------------------------------------
 * f16 metrics (half 2.7.1: IEEE binary16, RNE == F16C vcvtps2ph / vcvtph2ps)
 * ---------------------------------------------------------------------------------------- */
void qo_f32_to_f16(const float *in, uint16_t *out, size_t n) {
    for (size_t i = 0; i < n; i++) out[i] = _cvtss_sh(in[i], _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
}
void qo_f16_to_f32(const uint16_t *in, float *out, size_t n) {
    for (size_t i = 0; i < n; i++) out[i] = _cvtsh_ss(in[i]);
}
static inline float h2f(uint16_t h) { return _cvtsh_ss(h); }

/* 0 = dot, 1 = euclid, 2 = manhattan ; metric_f16/avx/{dot.rs:13-69, euclid.rs:13-75, manhattan.rs:13-77} */
static float half_avx(int op, const uint16_t *v1, const uint16_t *v2, size_t n) {
    const __m256 mask = _mm256_set1_ps(-0.0f);
    size_t m = n - (n % 32);
    const __m128i *p1 = (const __m128i *)v1, *p2 = (const __m128i *)v2;
    __m256 s[4] = {_mm256_setzero_ps(), _mm256_setzero_ps(), _mm256_setzero_ps(), _mm256_setzero_ps()};
    for (size_t i = 0; i < m; i += 32) {
        for (int r = 0; r < 4; r++) {
            __m256 a = _mm256_cvtph_ps(_mm_loadu_si128(p1 + r));
            __m256 b = _mm256_cvtph_ps(_mm_loadu_si128(p2 + r));
            if (op == 0) s[r] = _mm256_fmadd_ps(a, b, s[r]);
            else {
                __m256 d = _mm256_sub_ps(a, b);
                if (op == 1) s[r] = _mm256_fmadd_ps(d, d, s[r]);
                else s[r] = _mm256_add_ps(_mm256_andnot_ps(mask, d), s[r]);
            }
        }
        p1 += 4; p2 += 4;
    }
    const uint16_t *t1 = (const uint16_t *)p1, *t2 = (const uint16_t *)p2;
    float result = hsum256_ps_avx(s[0]) + hsum256_ps_avx(s[1]) + hsum256_ps_avx(s[2]) + hsum256_ps_avx(s[3]);
    for (size_t i = 0; i < n - m; i++) {
        float a = h2f(t1[i]), b = h2f(t2[i]);
        if (op == 0) result += a * b;
        else if (op == 1) { float d = a - b; result += d * d; }
        else result += fabsf(a - b);
    }
    return op == 0 ? result : -result;
}
/* metric_f16/sse/dot.rs:10-17 (and euclid/manhattan twins): convert whole vectors to f32,
 * then call the f32 SSE kernel */
static float half_sse(int op, const uint16_t *v1, const uint16_t *v2, size_t n) {
    float *a = (float *)malloc(sizeof(float) * (n ? n : 1)), *b = (float *)malloc(sizeof(float) * (n ? n : 1));
    qo_f16_to_f32(v1, a, n); qo_f16_to_f32(v2, b, n);
    float r = op == 0 ? dot_similarity_sse(a, b, n) : op == 1 ? euclid_similarity_sse(a, b, n) : manhattan_similarity_sse(a, b, n);
    free(a); free(b);
    return r;
}
/* metric_f16/simple_dot.rs:59-67, simple_euclid.rs:59-67, simple_manhattan.rs:59-67 */
static float half_scalar(int op, const uint16_t *v1, const uint16_t *v2, size_t n) {
    float s = RUST_SUM_INIT;
    for (size_t i = 0; i < n; i++) {
        float a = h2f(v1[i]), b = h2f(v2[i]);
        if (op == 0) s += a * b;
        else if (op == 1) { float d = a - b; s += d * d; }
        else s += fabsf(a - b);
    }
    return op == 0 ? s : -s;
}
static float half_dispatch(int op, const uint16_t *a, const uint16_t *b, size_t n, int isa) {
    switch (pick_isa(isa, n)) {
        case QO_ISA_AVX: return half_avx(op, a, b, n);
        case QO_ISA_SSE: return half_sse(op, a, b, n);
        default: return half_scalar(op, a, b, n);
    }
}
float qo_dot_f16(const uint16_t *a, const uint16_t *b, size_t n, int isa) { return half_dispatch(0, a, b, n, isa); }
float qo_euclid_f16(const uint16_t *a, const uint16_t *b, size_t n, int isa) { return half_dispatch(1, a, b, n, isa); }
float qo_manhattan_f16(const uint16_t *a, const uint16_t *b, size_t n, int isa) { return half_dispatch(2, a, b, n, isa); }
float qo_similarity_f16(int distance, const uint16_t *q, const uint16_t *v, size_t n) {
    switch (distance) {
        case QO_COSINE: /* metric_f16/simple_cosine.rs:28-58: dot on vectors normalised in f32 before the cast */
        case QO_DOT: return qo_dot_f16(q, v, n, QO_ISA_AUTO);
        case QO_EUCLID: return qo_euclid_f16(q, v, n, QO_ISA_AUTO);
        default: return qo_manhattan_f16(q, v, n, QO_ISA_AUTO);
    }
}

/* ------------------------------------------------------------------------------------------
 * u8 metrics
 * ---------------------------------------------------------------------------------------- */
void qo_f32_to_u8(const float *in, uint8_t *out, size_t n) {
    /* Rust `x as u8`: truncate toward zero, saturate, NaN -> 0 (primitive.rs:127-129) */
    for (size_t i = 0; i < n; i++) {
        float x = in[i];
        out[i] = (x != x) ? 0 : (x <= 0.0f ? 0 : (x >= 255.0f ? 255 : (uint8_t)x));
    }
}
/* metric_uint/avx2/dot.rs:9-69 */
static float avx_dot_similarity_bytes(const uint8_t *v1, const uint8_t *v2, size_t len) {
    const uint8_t *p1 = v1, *p2 = v2;
    __m256i dot_acc = _mm256_setzero_si256();
    const __m256i mask = _mm256_set1_epi16(0xFF);
    for (size_t i = 0; i < len / 32; i++) {
        __m256i a = _mm256_loadu_si256((const __m256i *)p1), b = _mm256_loadu_si256((const __m256i *)p2);
        p1 += 32; p2 += 32;
        __m256i a_lo = _mm256_and_si256(a, mask), a_hi = _mm256_and_si256(_mm256_bsrli_epi128(a, 1), mask);
        __m256i b_lo = _mm256_and_si256(b, mask), b_hi = _mm256_and_si256(_mm256_bsrli_epi128(b, 1), mask);
        dot_acc = _mm256_add_epi32(dot_acc, _mm256_madd_epi16(a_lo, b_lo));
        dot_acc = _mm256_add_epi32(dot_acc, _mm256_madd_epi16(a_hi, b_hi));
    }
    float score = hsum256_ps_avx(_mm256_cvtepi32_ps(dot_acc));
    size_t rem = len % 32;
    if (rem != 0) {
        int32_t r = 0;
        for (size_t i = 0; i < rem; i++) r += (int32_t)p1[i] * (int32_t)p2[i];
        score += (float)r;
    }
    return score;
}
/* metric_uint/avx2/cosine.rs:9-108 */
static float avx_cosine_similarity_bytes(const uint8_t *v1, const uint8_t *v2, size_t len) {
    const uint8_t *p1 = v1, *p2 = v2;
    __m256i dot_acc = _mm256_setzero_si256(), n1_acc = dot_acc, n2_acc = dot_acc;
    const __m256i mask = _mm256_set1_epi16(0xFF);
    for (size_t i = 0; i < len / 32; i++) {
        __m256i a = _mm256_loadu_si256((const __m256i *)p1), b = _mm256_loadu_si256((const __m256i *)p2);
        p1 += 32; p2 += 32;
        __m256i a_lo = _mm256_and_si256(a, mask), a_hi = _mm256_and_si256(_mm256_bsrli_epi128(a, 1), mask);
        __m256i b_lo = _mm256_and_si256(b, mask), b_hi = _mm256_and_si256(_mm256_bsrli_epi128(b, 1), mask);
        n1_acc = _mm256_add_epi32(n1_acc, _mm256_madd_epi16(a_lo, a_lo));
        n2_acc = _mm256_add_epi32(n2_acc, _mm256_madd_epi16(b_lo, b_lo));
        dot_acc = _mm256_add_epi32(dot_acc, _mm256_madd_epi16(a_lo, b_lo));
        n1_acc = _mm256_add_epi32(n1_acc, _mm256_madd_epi16(a_hi, a_hi));
        n2_acc = _mm256_add_epi32(n2_acc, _mm256_madd_epi16(b_hi, b_hi));
        dot_acc = _mm256_add_epi32(dot_acc, _mm256_madd_epi16(a_hi, b_hi));
    }
    float dot = hsum256_ps_avx(_mm256_cvtepi32_ps(dot_acc));
    float n1 = hsum256_ps_avx(_mm256_cvtepi32_ps(n1_acc));
    float n2 = hsum256_ps_avx(_mm256_cvtepi32_ps(n2_acc));
    size_t rem = len % 32;
    if (rem != 0) {
        int32_t rd = 0, r1 = 0, r2 = 0;
        for (size_t i = 0; i < rem; i++) {
            int32_t x = p1[i], y = p2[i];
            rd += x * y; r1 += x * x; r2 += y * y;
        }
        dot += (float)rd; n1 += (float)r1; n2 += (float)r2;
    }
    float denom = n1 * n2;
    if (denom == 0.0f) return 0.0f;
    return dot / sqrtf(denom);
}
/* metric_uint/avx2/euclid.rs:9-68 */
static float avx_euclid_similarity_bytes(const uint8_t *v1, const uint8_t *v2, size_t len) {
    const uint8_t *p1 = v1, *p2 = v2;
    __m256i acc = _mm256_setzero_si256();
    const __m256i mask = _mm256_set1_epi16(0xFF);
    for (size_t i = 0; i < len / 32; i++) {
        __m256i a = _mm256_loadu_si256((const __m256i *)p1), b = _mm256_loadu_si256((const __m256i *)p2);
        p1 += 32; p2 += 32;
        __m256i ad = _mm256_max_epu8(_mm256_subs_epu8(a, b), _mm256_subs_epu8(b, a));
        __m256i lo = _mm256_and_si256(ad, mask), hi = _mm256_and_si256(_mm256_bsrli_epi128(ad, 1), mask);
        acc = _mm256_add_epi32(acc, _mm256_madd_epi16(lo, lo));
        acc = _mm256_add_epi32(acc, _mm256_madd_epi16(hi, hi));
    }
    float score = hsum256_ps_avx(_mm256_cvtepi32_ps(acc));
    size_t rem = len % 32;
    if (rem != 0) {
        int32_t r = 0;
        for (size_t i = 0; i < rem; i++) { int32_t d = (int32_t)p1[i] - (int32_t)p2[i]; r += d * d; }
        score += (float)r;
    }
    return -score;
}
/* metric_uint/avx2/manhattan.rs:9-57 */
static float avx_manhattan_similarity_bytes(const uint8_t *v1, const uint8_t *v2, size_t len) {
    const uint8_t *p1 = v1, *p2 = v2;
    __m256i acc = _mm256_setzero_si256();
    for (size_t i = 0; i < len / 32; i++) {
        __m256i a = _mm256_loadu_si256((const __m256i *)p1), b = _mm256_loadu_si256((const __m256i *)p2);
        p1 += 32; p2 += 32;
        acc = _mm256_add_epi32(acc, _mm256_sad_epu8(a, b));
    }
    float score = hsum256_ps_avx(_mm256_cvtepi32_ps(acc));
    size_t rem = len % 32;
    if (rem != 0) {
        int32_t r = 0;
        for (size_t i = 0; i < rem; i++) r += abs((int32_t)p1[i] - (int32_t)p2[i]);
        score += (float)r;
    }
    return -score;
}
/* metric_uint/simple_{dot,cosine,euclid,manhattan}.rs — i32 exact, one cast */
static float scalar_bytes(int distance, const uint8_t *a, const uint8_t *b, size_t n) {
    int32_t dot = 0, n1 = 0, n2 = 0, l2 = 0, l1 = 0;
    for (size_t i = 0; i < n; i++) {
        int32_t x = a[i], y = b[i];
        dot += x * y; n1 += x * x; n2 += y * y;
        l2 += (x - y) * (x - y); l1 += abs(x - y);
    }
    switch (distance) {
        case QO_DOT: return (float)dot;                               /* simple_dot.rs:58-69 */
        case QO_COSINE:                                               /* simple_cosine.rs:58-77 */
            if (n1 == 0 || n2 == 0) return 0.0f;
            return (float)dot / sqrtf((float)n1 * (float)n2);
        case QO_EUCLID: return -(float)l2;                            /* simple_euclid.rs:58-69 */
        default: return -(float)l1;                                   /* simple_manhattan.rs:55-66 */
    }
}
/* metric_uint/sse2/{dot,cosine,euclid,manhattan}.rs — 4 i32 lanes over 16-byte steps, cvtepi32_ps, hsum128 */
static float sse_bytes(int distance, const uint8_t *v1, const uint8_t *v2, size_t len) {
    const uint8_t *p1 = v1, *p2 = v2;
    __m128i dot_acc = _mm_setzero_si128(), n1_acc = dot_acc, n2_acc = dot_acc, acc = dot_acc;
    const __m128i mask = _mm_set1_epi16(0xFF);
    for (size_t i = 0; i < len / 16; i++) {
        __m128i a = _mm_loadu_si128((const __m128i *)p1), b = _mm_loadu_si128((const __m128i *)p2);
        p1 += 16; p2 += 16;
        if (distance == QO_DOT || distance == QO_COSINE) {
            __m128i a_lo = _mm_and_si128(a, mask), a_hi = _mm_and_si128(_mm_bsrli_si128(a, 1), mask);
            __m128i b_lo = _mm_and_si128(b, mask), b_hi = _mm_and_si128(_mm_bsrli_si128(b, 1), mask);
            if (distance == QO_COSINE) {
                n1_acc = _mm_add_epi32(n1_acc, _mm_madd_epi16(a_lo, a_lo));
                n2_acc = _mm_add_epi32(n2_acc, _mm_madd_epi16(b_lo, b_lo));
            }
            dot_acc = _mm_add_epi32(dot_acc, _mm_madd_epi16(a_lo, b_lo));
            if (distance == QO_COSINE) {
                n1_acc = _mm_add_epi32(n1_acc, _mm_madd_epi16(a_hi, a_hi));
                n2_acc = _mm_add_epi32(n2_acc, _mm_madd_epi16(b_hi, b_hi));
            }
            dot_acc = _mm_add_epi32(dot_acc, _mm_madd_epi16(a_hi, b_hi));
        } else if (distance == QO_EUCLID) {
            __m128i ad = _mm_max_epu8(_mm_subs_epu8(a, b), _mm_subs_epu8(b, a));
            __m128i lo = _mm_and_si128(ad, mask), hi = _mm_and_si128(_mm_bsrli_si128(ad, 1), mask);
            acc = _mm_add_epi32(acc, _mm_madd_epi16(lo, lo));
            acc = _mm_add_epi32(acc, _mm_madd_epi16(hi, hi));
        } else {
            acc = _mm_add_epi32(acc, _mm_sad_epu8(a, b));
        }
    }
    size_t rem = len % 16;
    int32_t rd = 0, r1 = 0, r2 = 0, rl2 = 0, rl1 = 0;
    for (size_t i = 0; i < rem; i++) {
        int32_t x = p1[i], y = p2[i];
        rd += x * y; r1 += x * x; r2 += y * y; rl2 += (x - y) * (x - y); rl1 += abs(x - y);
    }
    if (distance == QO_DOT) {
        float s = hsum128_ps_sse(_mm_cvtepi32_ps(dot_acc));
        if (rem) s += (float)rd;
        return s;
    } else if (distance == QO_COSINE) {
        float d = hsum128_ps_sse(_mm_cvtepi32_ps(dot_acc));
        float n1 = hsum128_ps_sse(_mm_cvtepi32_ps(n1_acc)), n2 = hsum128_ps_sse(_mm_cvtepi32_ps(n2_acc));
        if (rem) { d += (float)rd; n1 += (float)r1; n2 += (float)r2; }
        float denom = n1 * n2;
        if (denom == 0.0f) return 0.0f;
        return d / sqrtf(denom);
    } else {
        float s = hsum128_ps_sse(_mm_cvtepi32_ps(acc));
        if (rem) s += (float)(distance == QO_EUCLID ? rl2 : rl1);
        return -s;
    }
}
float qo_similarity_u8(int distance, const uint8_t *q, const uint8_t *v, size_t n, int isa) {
    switch (pick_isa(isa, n)) {
        case QO_ISA_AVX:
            switch (distance) {
                case QO_DOT: return avx_dot_similarity_bytes(q, v, n);
                case QO_COSINE: return avx_cosine_similarity_bytes(q, v, n);
                case QO_EUCLID: return avx_euclid_similarity_bytes(q, v, n);
                default: return avx_manhattan_similarity_bytes(q, v, n);
            }
        case QO_ISA_SSE: return sse_bytes(distance, q, v, n);
        default: return scalar_bytes(distance, q, v, n);
    }
}
float qo_dot_u8(const uint8_t *a, const uint8_t *b, size_t n, int isa) { return qo_similarity_u8(QO_DOT, a, b, n, isa); }
float qo_cosine_u8(const uint8_t *a, const uint8_t *b, size_t n, int isa) { return qo_similarity_u8(QO_COSINE, a, b, n, isa); }
float qo_euclid_u8(const uint8_t *a, const uint8_t *b, size_t n, int isa) { return qo_similarity_u8(QO_EUCLID, a, b, n, isa); }
float qo_manhattan_u8(const uint8_t *a, const uint8_t *b, size_t n, int isa) { return qo_similarity_u8(QO_MANHATTAN, a, b, n, isa); }

/* ------------------------------------------------------------------------------------------
 * FixedLengthPriorityQueue<ScoredPointOffset> = BinaryHeap<Reverse<T>> + length
 * lib/common/common/src/fixed_length_priority_queue.rs:20-65.  BinaryHeap is Rust std
 * (alloc::collections::binary_heap, not in the qdrant tree): push = append + sift_up,
 * PeekMut drop = sift_down(0), into_sorted_vec = repeated swap(0,end) + sift_down_range(0,end).
 * ---------------------------------------------------------------------------------------- */
struct qo_topk { qo_scored_point *data; size_t len, length; };

/* OrderedFloat::cmp (ordered-float 5.3.0): NaN is greatest and equal to itself. types.rs:21-25 */
static inline int of_cmp(float a, float b) {
    if (a < b) return -1;
    if (a > b) return 1;
    if (a == b) return 0;
    int an = a != a, bn = b != b;
    if (an && bn) return 0;
    return an ? 1 : -1;
}
/* ordering of Reverse<ScoredPointOffset> */
static inline int rev_cmp(const qo_scored_point *a, const qo_scored_point *b) { return of_cmp(b->score, a->score); }

qo_topk *qo_topk_new(size_t length) {
    qo_topk *t = (qo_topk *)calloc(1, sizeof(*t));
    t->length = length ? length : 1;
    t->data = (qo_scored_point *)malloc(sizeof(qo_scored_point) * (t->length + 1));
    return t;
}
void qo_topk_free(qo_topk *t) { if (t) { free(t->data); free(t); } }

static void sift_up(qo_scored_point *d, size_t start, size_t pos) {
    qo_scored_point elt = d[pos];
    while (pos > start) {
        size_t parent = (pos - 1) / 2;
        if (rev_cmp(&elt, &d[parent]) <= 0) break;
        d[pos] = d[parent];
        pos = parent;
    }
    d[pos] = elt;
}
static void sift_down_range(qo_scored_point *d, size_t pos, size_t end) {
    qo_scored_point elt = d[pos];
    size_t child = 2 * pos + 1;
    size_t lim = end >= 2 ? end - 2 : 0; /* end.saturating_sub(2) */
    while (child <= lim && end >= 2) {
        child += (rev_cmp(&d[child], &d[child + 1]) <= 0) ? 1 : 0;
        if (rev_cmp(&elt, &d[child]) >= 0) { d[pos] = elt; return; }
        d[pos] = d[child];
        pos = child;
        child = 2 * pos + 1;
    }
    if (end >= 1 && child == end - 1 && rev_cmp(&elt, &d[child]) < 0) {
        d[pos] = d[child];
        pos = child;
    }
    d[pos] = elt;
}
void qo_topk_push(qo_topk *t, uint32_t idx, float score) {
    qo_scored_point v = {idx, score};
    if (t->len < t->length) {          /* !is_full(): heap.push */
        t->data[t->len] = v;
        sift_up(t->data, 0, t->len);
        t->len++;
        return;
    }
    /* full: replace the root (the smallest) only on strict root < value (:53-57) */
    if (of_cmp(t->data[0].score, v.score) < 0) {
        t->data[0] = v;
        sift_down_range(t->data, 0, t->len);
    }
}
/* push with the Option<T> result of FixedLengthPriorityQueue::push (:47-59):
 * 0 = None (queue was not full), 1 = Some(old root) written to *removed (value entered),
 * 2 = Some(value) (value rejected; *removed = value). */
int qo_topk_push_ex(qo_topk *t, uint32_t idx, float score, qo_scored_point *removed) {
    qo_scored_point v = {idx, score};
    if (t->len < t->length) {
        t->data[t->len] = v;
        sift_up(t->data, 0, t->len);
        t->len++;
        return 0;
    }
    if (of_cmp(t->data[0].score, v.score) < 0) {
        if (removed) *removed = t->data[0];
        t->data[0] = v;
        sift_down_range(t->data, 0, t->len);
        return 1;
    }
    if (removed) *removed = v;
    return 2;
}
/* top(): the smallest element kept (heap.peek()), :80-82 */
int qo_topk_top(const qo_topk *t, qo_scored_point *out) {
    if (t->len == 0) return 0;
    *out = t->data[0];
    return 1;
}
size_t qo_topk_len(const qo_topk *t) { return t->len; }
/* iter_unsorted(): heap storage order (:68-70) */
const qo_scored_point *qo_topk_data(const qo_topk *t) { return t->data; }
int qo_ordered_float_cmp(float a, float b) { return of_cmp(a, b); }

size_t qo_topk_into_sorted(qo_topk *t, qo_scored_point *out) {
    size_t end = t->len;
    while (end > 1) {
        end--;
        qo_scored_point tmp = t->data[0]; t->data[0] = t->data[end]; t->data[end] = tmp;
        sift_down_range(t->data, 0, end);
    }
    /* ascending in Reverse<T> order == descending score */
    memcpy(out, t->data, sizeof(qo_scored_point) * t->len);
    size_t n = t->len;
    t->len = 0;
    return n;
}

/* ------------------------------------------------------------------------------------------
 * brute-force search
 * ---------------------------------------------------------------------------------------- */
static inline size_t elem_size(int dtype) { return dtype == QO_F32 ? 4 : dtype == QO_F16 ? 2 : 1; }
static inline int get_bit(const uint64_t *bits, size_t i) { return (int)((bits[i >> 6] >> (i & 63)) & 1); } /* BitSlice<u64, Lsb0> */

/* NotDeletedChecker::check (raw_scorer.rs:596-603) */
static inline int check_vector(const qo_storage *st, uint32_t id) {
    int vec_del = (st->vec_deleted && id < st->n_vec_bits) ? get_bit(st->vec_deleted, id) : 0;
    int pt_del;
    if (st->point_deleted) pt_del = id < st->n_point_bits ? get_bit(st->point_deleted, id) : 1;
    else pt_del = id < st->n ? 0 : 1;
    return !vec_del && !pt_del;
}
static inline float score_one(const qo_storage *st, const void *q, uint32_t id) {
    const char *row = (const char *)st->rows + (size_t)id * st->dim * elem_size(st->dtype);
    switch (st->dtype) {
        case QO_F32: return qo_similarity_f32(st->distance, (const float *)q, (const float *)row, st->dim);
        case QO_F16: return qo_similarity_f16(st->distance, (const uint16_t *)q, (const uint16_t *)row, st->dim);
        default: return qo_similarity_u8(st->distance, (const uint8_t *)q, (const uint8_t *)row, st->dim, st->u8_isa);
    }
}
void qo_score_points(const qo_storage *st, const void *query, const uint32_t *ids, size_t n, float *out) {
    for (size_t i = 0; i < n; i++) out[i] = score_one(st, query, ids[i]);
}

#define VECTOR_READ_BATCH_SIZE 64 /* vector_storage/common.rs:20 */

static int peek_range(const qo_storage *st, const void *queries, size_t nq, qo_topk **pqs,
                      const uint32_t *ids, size_t n_ids, size_t lo, size_t hi,
                      const volatile uint8_t *is_stopped) {
    uint32_t chunk[VECTOR_READ_BATCH_SIZE];
    float scores[VECTOR_READ_BATCH_SIZE];
    size_t qstride = st->dim * elem_size(st->dtype);
    size_t cursor = lo;
    size_t end = ids ? n_ids : hi;
    if (ids) { cursor = 0; }
    for (;;) {
        if (is_stopped && *is_stopped) return 7;
        size_t chunk_size = 0;
        while (cursor < end) {
            uint32_t pid;
            if (ids) pid = ids[cursor++];
            else {
                /* iter_zeros of point_deleted (point_scorer.rs:400-406) */
                pid = (uint32_t)cursor++;
                if (st->point_deleted && get_bit(st->point_deleted, pid)) continue;
            }
            if (!check_vector(st, pid)) continue;
            chunk[chunk_size++] = pid;
            if (chunk_size == VECTOR_READ_BATCH_SIZE) break;
        }
        if (chunk_size == 0) break;
        for (size_t qi = 0; qi < nq; qi++) {
            const void *q = (const char *)queries + qi * qstride;
            for (size_t i = 0; i < chunk_size; i++) scores[i] = score_one(st, q, chunk[i]);
            for (size_t i = 0; i < chunk_size; i++) qo_topk_push(pqs[qi], chunk[i], scores[i]);
        }
    }
    return 0;
}

int qo_peek_top_iter(const qo_storage *st, const void *queries, size_t nq, size_t top,
                     const uint32_t *ids, size_t n_ids, qo_scored_point *out, uint32_t *counts,
                     const volatile uint8_t *is_stopped) {
    qo_topk **pqs = (qo_topk **)malloc(sizeof(*pqs) * (nq ? nq : 1));
    for (size_t i = 0; i < nq; i++) pqs[i] = qo_topk_new(top);
    size_t hi = st->point_deleted ? (st->n_point_bits < st->n ? st->n_point_bits : st->n) : st->n;
    int rc = peek_range(st, queries, nq, pqs, ids, n_ids, 0, hi, is_stopped);
    for (size_t i = 0; i < nq; i++) {
        if (rc == 0) counts[i] = (uint32_t)qo_topk_into_sorted(pqs[i], out + i * top);
        qo_topk_free(pqs[i]);
    }
    free(pqs);
    return rc;
}

typedef struct {
    const qo_storage *st; const void *queries; size_t nq, top, lo, hi; qo_topk **pqs;
} par_arg;
static void *par_worker(void *p) {
    par_arg *a = (par_arg *)p;
    peek_range(a->st, a->queries, a->nq, a->pqs, NULL, 0, a->lo, a->hi, NULL);
    return NULL;
}
int qo_peek_top_parallel(const qo_storage *st, const void *queries, size_t nq, size_t top,
                         qo_scored_point *out, uint32_t *counts, int threads) {
    if (threads < 1) threads = 1;
    size_t hi = st->point_deleted ? (st->n_point_bits < st->n ? st->n_point_bits : st->n) : st->n;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * threads);
    par_arg *args = (par_arg *)malloc(sizeof(par_arg) * threads);
    for (int t = 0; t < threads; t++) {
        args[t].st = st; args[t].queries = queries; args[t].nq = nq; args[t].top = top;
        args[t].lo = hi * t / threads; args[t].hi = hi * (t + 1) / threads;
        args[t].pqs = (qo_topk **)malloc(sizeof(qo_topk *) * nq);
        for (size_t i = 0; i < nq; i++) args[t].pqs[i] = qo_topk_new(top);
        pthread_create(&th[t], NULL, par_worker, &args[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    /* merge like BatchResultAggregator (search_result_aggregator.rs:50-121): push every partial
     * result into one bounded queue per query */
    qo_scored_point *tmp = (qo_scored_point *)malloc(sizeof(qo_scored_point) * (top ? top : 1));
    for (size_t qi = 0; qi < nq; qi++) {
        qo_topk *m = qo_topk_new(top);
        for (int t = 0; t < threads; t++) {
            size_t c = qo_topk_into_sorted(args[t].pqs[qi], tmp);
            for (size_t i = 0; i < c; i++) qo_topk_push(m, tmp[i].idx, tmp[i].score);
        }
        counts[qi] = (uint32_t)qo_topk_into_sorted(m, out + qi * top);
        qo_topk_free(m);
    }
    free(tmp);
    for (int t = 0; t < threads; t++) { for (size_t i = 0; i < nq; i++) qo_topk_free(args[t].pqs[i]); free(args[t].pqs); }
    free(args); free(th);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * SQ int8 — lib/quantization/src/encoded_vectors_u8.rs
 * ---------------------------------------------------------------------------------------- */
#define SQ_ALIGNMENT 16
static uint32_t sq_actual_dim(uint32_t dim) { return dim + (SQ_ALIGNMENT - dim % SQ_ALIGNMENT) % SQ_ALIGNMENT; } /* :622-624 */
static int sq_is_dot(int distance) { return distance == QO_DOT || distance == QO_COSINE; }

void qo_sq_init_params(qo_sq *sq, int distance, int invert, uint32_t dim, float alpha, float offset) {
    sq->dim = dim; sq->actual_dim = sq_actual_dim(dim);
    sq->distance = distance; sq->invert = invert; sq->alpha = alpha; sq->offset = offset;
    float multiplier;                         /* :205-221 */
    if (sq_is_dot(distance)) multiplier = alpha * alpha;
    else if (distance == QO_MANHATTAN) multiplier = alpha;
    else multiplier = -2.0f * alpha * alpha;
    sq->multiplier = invert ? -multiplier : multiplier;
}
void qo_sq_init(qo_sq *sq, int distance, int invert, uint32_t dim, const float *data, size_t n) {
    /* find_min_max_from_iter (quantile.rs) + alpha_offset_from_min_max (:529-533) */
    float mn = INFINITY, mx = -INFINITY;
    for (size_t i = 0; i < n * (size_t)dim; i++) { if (data[i] < mn) mn = data[i]; if (data[i] > mx) mx = data[i]; }
    float alpha = (mx - mn) / 127.0f;
    qo_sq_init_params(sq, distance, invert, dim, alpha, mn);
}
uint8_t qo_sq_encode_value(const qo_sq *sq, float value) {
    float i = (value - sq->offset) / sq->alpha;           /* :94-98 */
    /* f32::clamp(0,127): NaN stays NaN; then round() half away from zero; `as u8` NaN -> 0 */
    if (i != i) return 0;
    if (i < 0.0f) i = 0.0f;
    if (i > 127.0f) i = 127.0f;
    return (uint8_t)roundf(i);
}
float qo_sq_get_shift(const qo_sq *sq) {                  /* :116-134 */
    float shift = sq_is_dot(sq->distance) ? (float)sq->actual_dim * sq->offset * sq->offset : 0.0f;
    return sq->invert ? -shift : shift;
}
static float sq_codes_offset(const qo_sq *sq, const uint8_t *codes, int with_header_zeros) {
    /* sequential f32 sums exactly as :258-275 / :596-609 (header zeros add nothing) */
    (void)with_header_zeros;
    float off;
    if (sq_is_dot(sq->distance)) {
        float s = RUST_SUM_INIT;
        for (uint32_t i = 0; i < sq->actual_dim; i++) s += (float)codes[i];
        off = s * sq->alpha * sq->offset;
    } else if (sq->distance == QO_MANHATTAN) {
        off = 0.0f;
    } else {
        float s = RUST_SUM_INIT;
        for (uint32_t i = 0; i < sq->actual_dim; i++) s += (float)codes[i] * (float)codes[i];
        off = s * sq->alpha * sq->alpha;
    }
    return sq->invert ? -off : off;
}
static void sq_encode_codes(const qo_sq *sq, const float *v, uint8_t *codes) {
    for (uint32_t i = 0; i < sq->dim; i++) codes[i] = qo_sq_encode_value(sq, v[i]);
    float placeholder = sq_is_dot(sq->distance) ? 0.0f : sq->offset;       /* :246-255, :586-595 */
    for (uint32_t i = sq->dim; i < sq->actual_dim; i++) codes[i] = qo_sq_encode_value(sq, placeholder);
}
void qo_sq_encode_row(const qo_sq *sq, const float *v, uint8_t *out_row) {
    uint8_t *codes = out_row + 4;
    sq_encode_codes(sq, v, codes);
    float vector_offset = sq_codes_offset(sq, codes, 1);
    vector_offset = qo_sq_get_shift(sq) + vector_offset;                  /* :281-283 */
    memcpy(out_row, &vector_offset, 4);
}
void qo_sq_encode_query(const qo_sq *sq, const float *q, uint8_t *codes, float *q_offset) {
    sq_encode_codes(sq, q, codes);
    *q_offset = sq_codes_offset(sq, codes, 0);
}
/* :813-831 scalar leaves */
static int32_t impl_score_dot(const uint8_t *q, const uint8_t *v, size_t n) {
    int32_t s = 0; for (size_t i = 0; i < n; i++) s += (int32_t)q[i] * (int32_t)v[i]; return s;
}
static int32_t impl_score_l1(const uint8_t *q, const uint8_t *v, size_t n) {
    int32_t s = 0; for (size_t i = 0; i < n; i++) s += abs((int32_t)q[i] - (int32_t)v[i]); return s;
}
/* cpp/avx2.c:25-63 */
float qo_sq_dot_avx(const uint8_t *query_ptr, const uint8_t *vector_ptr, uint32_t dim) {
    const __m256i *v_ptr = (const __m256i *)vector_ptr, *q_ptr = (const __m256i *)query_ptr;
    __m256i mul1 = _mm256_setzero_si256();
    __m256i mask_epu32 = _mm256_set1_epi32(0xFFFF);
    for (uint32_t i = 0; i < dim / 32; i++) {
        __m256i v = _mm256_loadu_si256(v_ptr++), q = _mm256_loadu_si256(q_ptr++);
        __m256i s = _mm256_maddubs_epi16(v, q);
        mul1 = _mm256_add_epi32(mul1, _mm256_cvtepi16_epi32(_mm256_castsi256_si128(s)));
        mul1 = _mm256_add_epi32(mul1, _mm256_cvtepi16_epi32(_mm256_extractf128_si256(s, 1)));
    }
    if (dim % 32 != 0) {
        __m256i v1 = _mm256_cvtepu8_epi16(_mm_loadu_si128((const __m128i *)v_ptr));
        __m256i q1 = _mm256_cvtepu8_epi16(_mm_loadu_si128((const __m128i *)q_ptr));
        __m256i s = _mm256_mullo_epi16(v1, q1);
        mul1 = _mm256_add_epi32(mul1, _mm256_and_si256(s, mask_epu32));
        mul1 = _mm256_add_epi32(mul1, _mm256_srli_epi32(s, 16));
    }
    __m256 mul_ps = _mm256_cvtepi32_ps(mul1);
    __m128 x128 = _mm_add_ps(_mm256_extractf128_ps(mul_ps, 1), _mm256_castps256_ps128(mul_ps));
    __m128 x64 = _mm_add_ps(x128, _mm_movehl_ps(x128, x128));
    __m128 x32 = _mm_add_ss(x64, _mm_shuffle_ps(x64, x64, 0x55));
    return _mm_cvtss_f32(x32);
}
/* cpp/avx2.c:65-122 */
float qo_sq_l1_avx(const uint8_t *query_ptr, const uint8_t *vector_ptr, uint32_t dim) {
    const __m256i *v_ptr = (const __m256i *)vector_ptr, *q_ptr = (const __m256i *)query_ptr;
    uint32_t m = dim - (dim % 32);
    __m256i sum256 = _mm256_setzero_si256();
    for (uint32_t i = 0; i < m; i += 32) {
        __m256i v = _mm256_loadu_si256(v_ptr++), q = _mm256_loadu_si256(q_ptr++);
        __m256i ad = _mm256_max_epu8(_mm256_subs_epu8(v, q), _mm256_subs_epu8(q, v));
        sum256 = _mm256_add_epi16(sum256, _mm256_unpacklo_epi8(ad, _mm256_setzero_si256()));
        sum256 = _mm256_add_epi16(sum256, _mm256_unpackhi_epi8(ad, _mm256_setzero_si256()));
    }
    if (m < dim) {
        __m128i vs = _mm_loadu_si128((const __m128i *)v_ptr), qs = _mm_loadu_si128((const __m128i *)q_ptr);
        __m128i ad = _mm_max_epu8(_mm_subs_epu8(vs, qs), _mm_subs_epu8(qs, vs));
        __m256i lo = _mm256_cvtepu16_epi32(_mm_unpacklo_epi8(ad, _mm_setzero_si128()));
        __m256i hi = _mm256_cvtepu16_epi32(_mm_unpackhi_epi8(ad, _mm_setzero_si128()));
        sum256 = _mm256_add_epi16(sum256, lo);
        sum256 = _mm256_add_epi16(sum256, hi);
    }
    __m256i sum_epi32 = _mm256_add_epi32(_mm256_unpacklo_epi16(sum256, _mm256_setzero_si256()),
                                         _mm256_unpackhi_epi16(sum256, _mm256_setzero_si256()));
    __m128i x128 = _mm_add_epi32(_mm256_extractf128_si256(sum_epi32, 1), _mm256_castsi256_si128(sum_epi32));
    __m128i x64 = _mm_add_epi32(x128, _mm_srli_si128(x128, 8));
    __m128i x32 = _mm_add_epi32(x64, _mm_srli_si128(x64, 4));
    return (float)_mm_cvtsi128_si32(x32);
}
/* cpp/sse.c:28-52 */
float qo_sq_dot_sse(const uint8_t *query_ptr, const uint8_t *vector_ptr, uint32_t dim) {
    const __m128i *v_ptr = (const __m128i *)vector_ptr, *q_ptr = (const __m128i *)query_ptr;
    __m128i mul = _mm_setzero_si128();
    for (uint32_t i = 0; i < dim / 16; i++) {
        __m128i v = _mm_loadu_si128(v_ptr++), q = _mm_loadu_si128(q_ptr++);
        __m128i s = _mm_maddubs_epi16(v, q);
        mul = _mm_add_epi32(mul, _mm_cvtepi16_epi32(s));
        mul = _mm_add_epi32(mul, _mm_cvtepi16_epi32(_mm_srli_si128(s, 8)));
    }
    return hsum128_ps_sse(_mm_cvtepi32_ps(mul));
}
/* cpp/sse.c:472-513 (including its 16-bit horizontal add, HSUM128_EPI16 :20-26) */
float qo_sq_l1_sse(const uint8_t *query_ptr, const uint8_t *vector_ptr, uint32_t dim) {
    const __m128i *v_ptr = (const __m128i *)vector_ptr, *q_ptr = (const __m128i *)query_ptr;
    uint32_t m = dim - (dim % 16);
    __m128i sum128 = _mm_setzero_si128();
    for (uint32_t i = 0; i < m; i += 16) {
        __m128i vec2 = _mm_loadu_si128(v_ptr++), vec1 = _mm_loadu_si128(q_ptr++);
        __m128i ad = _mm_max_epu8(_mm_subs_epu8(vec1, vec2), _mm_subs_epu8(vec2, vec1));
        sum128 = _mm_add_epi16(sum128, _mm_unpacklo_epi8(ad, _mm_setzero_si128()));
        sum128 = _mm_add_epi16(sum128, _mm_unpackhi_epi8(ad, _mm_setzero_si128()));
    }
    __m128i sum_epi32 = _mm_add_epi32(_mm_unpacklo_epi16(sum128, _mm_setzero_si128()),
                                      _mm_unpackhi_epi16(sum128, _mm_setzero_si128()));
    __m128i x64 = _mm_add_epi16(sum_epi32, _mm_srli_si128(sum_epi32, 8));
    __m128i x32 = _mm_add_epi16(x64, _mm_srli_si128(x64, 4));
    int sum = _mm_extract_epi16(x32, 0) + _mm_extract_epi16(x32, 1);
    return (float)sum;
}
static qo_sq_leaf_fn g_ref_dot_avx, g_ref_l1_avx;
void qo_sq_set_ref_kernels(qo_sq_leaf_fn dot_avx, qo_sq_leaf_fn l1_avx) { g_ref_dot_avx = dot_avx; g_ref_l1_avx = l1_avx; }

static float sq_leaf(const qo_sq *sq, const uint8_t *q, const uint8_t *v, int isa) {
    int l1 = sq->distance == QO_MANHATTAN;   /* Dot | Cosine | L2 use the dot leaf (:483-490) */
    switch (isa) {
        case QO_ISA_SCALAR: return l1 ? (float)impl_score_l1(q, v, sq->actual_dim) : (float)impl_score_dot(q, v, sq->actual_dim);
        case QO_ISA_SSE: return l1 ? qo_sq_l1_sse(q, v, sq->actual_dim) : qo_sq_dot_sse(q, v, sq->actual_dim);
        case 100: /* the reference's own C kernel from oracle/_ref */
            return l1 ? g_ref_l1_avx(q, v, sq->actual_dim) : g_ref_dot_avx(q, v, sq->actual_dim);
        default: return l1 ? qo_sq_l1_avx(q, v, sq->actual_dim) : qo_sq_dot_avx(q, v, sq->actual_dim);
    }
}
float qo_sq_score(const qo_sq *sq, const uint8_t *q_codes, float q_offset, const uint8_t *row, int isa) {
    float vector_offset; memcpy(&vector_offset, row, 4);                 /* parse_vec_data :542-550 */
    float score = sq_leaf(sq, q_codes, row + 4, isa);
    return sq->multiplier * score + q_offset + vector_offset;            /* postprocess_score :100-103 */
}
float qo_sq_score_internal(const qo_sq *sq, const uint8_t *row_i, const uint8_t *row_j, int isa) {
    float off_i, off_j; memcpy(&off_i, row_i, 4); memcpy(&off_j, row_j, 4);
    float score = sq_leaf(sq, row_i + 4, row_j + 4, isa);
    float query_offset = off_i - qo_sq_get_shift(sq);                    /* postprocess_internal_score :105-114 */
    return sq->multiplier * score + query_offset + off_j;
}

/* ------------------------------------------------------------------------------------------
 * PQ — lib/quantization/src/encoded_vectors_pq.rs
 * ---------------------------------------------------------------------------------------- */
void qo_pq_init(qo_pq *pq, int distance, int invert, uint32_t dim, uint32_t chunk_size,
                uint32_t n_centroids, const float *centroids) {
    pq->dim = dim; pq->chunk_size = chunk_size; pq->m = (dim + chunk_size - 1) / chunk_size; /* :164-169 */
    pq->n_centroids = n_centroids; pq->distance = distance; pq->invert = invert; pq->centroids = centroids;
}
/* DistanceType::distance (encoded_vectors.rs:119-127): sequential sum, mul and add un-fused */
static float pq_distance(int distance, const float *a, const float *b, size_t n) {
    float s = RUST_SUM_INIT;
    if (distance == QO_DOT || distance == QO_COSINE) for (size_t i = 0; i < n; i++) s += a[i] * b[i];
    else if (distance == QO_MANHATTAN) for (size_t i = 0; i < n; i++) s += fabsf(a[i] - b[i]);
    else for (size_t i = 0; i < n; i++) { float d = a[i] - b[i]; s += d * d; }
    return s;
}
void qo_pq_encode_vector(const qo_pq *pq, const float *v, uint8_t *codes) {
    for (uint32_t c = 0; c < pq->m; c++) {
        uint32_t lo = c * pq->chunk_size, hi = lo + pq->chunk_size; if (hi > pq->dim) hi = pq->dim;
        float min_distance = 3.40282347e+38f; /* f32::MAX */
        uint32_t min_idx = 0;
        for (uint32_t j = 0; j < pq->n_centroids; j++) {
            const float *cen = pq->centroids + (size_t)j * pq->dim;
            float s = RUST_SUM_INIT;
            for (uint32_t i = lo; i < hi; i++) { float d = v[i] - cen[i]; s += d * d; }
            if (s < min_distance) { min_distance = s; min_idx = j; }     /* first minimum wins (:321) */
        }
        codes[c] = (uint8_t)min_idx;
    }
}
void qo_pq_encode_query(const qo_pq *pq, const float *q, float *lut) {
    for (uint32_t c = 0; c < pq->m; c++) {
        uint32_t lo = c * pq->chunk_size, hi = lo + pq->chunk_size; if (hi > pq->dim) hi = pq->dim;
        for (uint32_t j = 0; j < pq->n_centroids; j++) {
            float d = pq_distance(pq->distance, q + lo, pq->centroids + (size_t)j * pq->dim + lo, hi - lo);
            lut[(size_t)c * pq->n_centroids + j] = pq->invert ? -d : d;
        }
    }
}
float qo_pq_score(const qo_pq *pq, const float *lut, const uint8_t *codes, int isa) {
    size_t len = pq->m, cc = pq->n_centroids;
    if (isa == QO_ISA_SCALAR) {                                            /* :478-493 */
        float s = RUST_SUM_INIT;
        for (size_t i = 0; i < len; i++) s += lut[i * cc + codes[i]];
        return s;
    }
    /* score_point_sse :409-443 */
    const uint8_t *c = codes; const float *l = lut;
    __m128 sum128 = _mm_setzero_ps();
    for (size_t i = 0; i < len / 4; i++) {
        float buffer[4] = { l[c[0]], l[cc + c[1]], l[2 * cc + c[2]], l[3 * cc + c[3]] };
        sum128 = _mm_add_ps(sum128, _mm_loadu_ps(buffer));
        c += 4; l += 4 * cc;
    }
    __m128 sum64 = _mm_add_ps(sum128, _mm_movehl_ps(sum128, sum128));
    __m128 sum32 = _mm_add_ss(sum64, _mm_shuffle_ps(sum64, sum64, 0x55));
    float sum = _mm_cvtss_f32(sum32);
    for (size_t i = 0; i < len % 4; i++) { sum += l[*c]; c++; l += cc; }
    return sum;
}
float qo_pq_score_internal(const qo_pq *pq, const uint8_t *ci, const uint8_t *cj) {
    float s = RUST_SUM_INIT;
    for (uint32_t c = 0; c < pq->m; c++) {
        uint32_t lo = c * pq->chunk_size, hi = lo + pq->chunk_size; if (hi > pq->dim) hi = pq->dim;
        s += pq_distance(pq->distance, pq->centroids + (size_t)ci[c] * pq->dim + lo,
                         pq->centroids + (size_t)cj[c] * pq->dim + lo, hi - lo);
    }
    return pq->invert ? -s : s;
}
/* Lloyd with first-k init (kmeans.rs:27), f64 accumulators (:76-110); empty clusters keep their
 * previous centroid here (the reference re-seeds them randomly, :113-120). */
void qo_pq_train(uint32_t dim, uint32_t chunk_size, uint32_t n_centroids, const float *data,
                 size_t n, int iters, float *centroids_out) {
    uint32_t m = (dim + chunk_size - 1) / chunk_size;
    if (n <= n_centroids) {  /* encoded_vectors_pq.rs:354-362 */
        memset(centroids_out, 0, sizeof(float) * (size_t)n_centroids * dim);
        for (size_t i = 0; i < n; i++) memcpy(centroids_out + i * dim, data + i * dim, sizeof(float) * dim);
        return;
    }
    double *acc = (double *)malloc(sizeof(double) * n_centroids * chunk_size);
    size_t *cnt = (size_t *)malloc(sizeof(size_t) * n_centroids);
    for (uint32_t c = 0; c < m; c++) {
        uint32_t lo = c * chunk_size, hi = lo + chunk_size; if (hi > dim) hi = dim;
        uint32_t w = hi - lo;
        for (uint32_t j = 0; j < n_centroids; j++)
            memcpy(centroids_out + (size_t)j * dim + lo, data + (size_t)j * dim + lo, sizeof(float) * w);
        for (int it = 0; it < iters; it++) {
            memset(acc, 0, sizeof(double) * n_centroids * chunk_size);
            memset(cnt, 0, sizeof(size_t) * n_centroids);
            for (size_t r = 0; r < n; r++) {
                const float *v = data + r * dim + lo;
                float best = 3.40282347e+38f; uint32_t bj = 0;
                for (uint32_t j = 0; j < n_centroids; j++) {
                    const float *cen = centroids_out + (size_t)j * dim + lo;
                    float s = 0.0f;
                    for (uint32_t i = 0; i < w; i++) { float d = v[i] - cen[i]; s += d * d; }
                    if (s < best) { best = s; bj = j; }
                }
                cnt[bj]++;
                for (uint32_t i = 0; i < w; i++) acc[(size_t)bj * chunk_size + i] += v[i];
            }
            for (uint32_t j = 0; j < n_centroids; j++) if (cnt[j])
                for (uint32_t i = 0; i < w; i++)
                    centroids_out[(size_t)j * dim + lo + i] = (float)(acc[(size_t)j * chunk_size + i] / (double)cnt[j]);
        }
    }
    free(acc); free(cnt);
}

/* kmeans (lib/quantization/src/kmeans.rs:9-169) on a GIVEN sample, per chunk as find_centroids calls it
 * (encoded_vectors_pq.rs:342-407): first-k init (:27), update_indexes (:139-169: f32 sum of (a-b)^2, first minimum),
 * update_centroids (:51-137): `threads` row ranges, each accumulated in f64 in row order, partials added in thread
 * order, mean cast to f32, stop when sum(|old - new|) (f32, index order) < accuracy or after max_iters.
 * Unpinned in the reference: the sample (Permutor) and the random re-seed of EMPTY clusters (:113-120) — an empty
 * cluster keeps its previous centroid here.  iters_done[c] (optional) = update steps executed for chunk c. */
void qo_pq_train_ex(uint32_t dim, uint32_t chunk_size, uint32_t n_centroids, const float *data, size_t n, uint32_t max_iters,
                    float accuracy, uint32_t threads, float *centroids_out, uint32_t *iters_done) {
    const uint32_t m = (dim + chunk_size - 1) / chunk_size;
    if (threads == 0) threads = 1;
    if (n <= n_centroids) {  /* encoded_vectors_pq.rs:354-362 */
        memset(centroids_out, 0, sizeof(float) * (size_t)n_centroids * dim);
        for (size_t i = 0; i < n; i++) memcpy(centroids_out + i * dim, data + i * dim, sizeof(float) * dim);
        if (iters_done) for (uint32_t c = 0; c < m; c++) iters_done[c] = 0;
        return;
    }
    double *acc = (double *)malloc(sizeof(double) * n_centroids * chunk_size);
    double *part = (double *)malloc(sizeof(double) * n_centroids * chunk_size);
    size_t *cnt = (size_t *)malloc(sizeof(size_t) * n_centroids);
    uint32_t *idx = (uint32_t *)malloc(sizeof(uint32_t) * n);
    for (uint32_t c = 0; c < m; c++) {
        uint32_t lo = c * chunk_size, hi = lo + chunk_size; if (hi > dim) hi = dim;
        const uint32_t w = hi - lo;
        for (uint32_t j = 0; j < n_centroids; j++)
            memcpy(centroids_out + (size_t)j * dim + lo, data + (size_t)j * dim + lo, sizeof(float) * w);
        uint32_t it = 0;
        for (; it < max_iters; ) {
            for (size_t r = 0; r < n; r++) {                      /* update_indexes */
                const float *v = data + r * dim + lo;
                float best = 3.40282347e+38f; uint32_t bj = 0;
                for (uint32_t j = 0; j < n_centroids; j++) {
                    const float *cen = centroids_out + (size_t)j * dim + lo;
                    float sm = -0.0f;
                    for (uint32_t i = 0; i < w; i++) { const float d = v[i] - cen[i]; sm += d * d; }
                    if (sm < best) { best = sm; bj = j; }
                }
                idx[r] = bj;
            }
            memset(acc, 0, sizeof(double) * n_centroids * chunk_size);
            memset(cnt, 0, sizeof(size_t) * n_centroids);
            const size_t per = n / threads;
            for (uint32_t t = 0; t < threads; t++) {              /* one CentroidsCounter per thread, summed in order */
                const size_t r0 = per * t, r1 = (t + 1 == threads) ? n : per * (t + 1);
                memset(part, 0, sizeof(double) * n_centroids * chunk_size);
                for (size_t r = r0; r < r1; r++) {
                    const float *v = data + r * dim + lo;
                    cnt[idx[r]]++;
                    for (uint32_t i = 0; i < w; i++) part[(size_t)idx[r] * chunk_size + i] += (double)v[i];
                }
                for (size_t k = 0; k < (size_t)n_centroids * chunk_size; k++) acc[k] += part[k];
            }
            float diff = -0.0f;
            for (uint32_t j = 0; j < n_centroids; j++)
                for (uint32_t i = 0; i < w; i++) {
                    float *cp = &centroids_out[(size_t)j * dim + lo + i];
                    const float nv = cnt[j] ? (float)(acc[(size_t)j * chunk_size + i] / (double)cnt[j]) : *cp;
                    diff += fabsf(*cp - nv);
                    *cp = nv;
                }
            it++;
            if (diff < accuracy) break;
        }
        if (iters_done) iters_done[c] = it;
    }
    free(acc); free(part); free(cnt); free(idx);
}

/* ------------------------------------------------------------------------------------------
 * custom queries: Query::score_by of RecoBestScoreQuery (vector_storage/query/reco_query.rs:68-92),
 * RecoSumScoresQuery (:114-131), DiscoverQuery (discover_query.rs:45-73, ContextPair::rank_by context_query.rs:38-45),
 * ContextQuery (context_query.rs:53-62, 112-118); fast_sigmoid / scaled_fast_sigmoid lib/common/common/src/math.rs:7-18.
 * sims = similarity(example, point) in flat_iter() order.  kind: 0 best score, 1 sum scores, 2 discover, 3 context.
 * ---------------------------------------------------------------------------------------- */
static int f32_total_cmp(float a, float b) {
    int32_t x, y;
    memcpy(&x, &a, 4); memcpy(&y, &b, 4);
    x ^= (int32_t)(((uint32_t)(x >> 31)) >> 1);
    y ^= (int32_t)(((uint32_t)(y >> 31)) >> 1);
    return x < y ? -1 : x > y ? 1 : 0;
}
static float fast_sigmoid(float x) { return x / (1.0f + fabsf(x)); }
static float scaled_fast_sigmoid(float x) { return 0.5f * (fast_sigmoid(x) + 1.0f); }
float qo_custom_combine(int kind, uint32_t n_a, uint32_t n_b, const float *sims) {
    if (kind == 0) {
        float max_pos = -INFINITY, max_neg = -INFINITY;
        for (uint32_t i = 0; i < n_a; i++) if (f32_total_cmp(sims[i], max_pos) > 0) max_pos = sims[i];
        for (uint32_t i = 0; i < n_b; i++) if (f32_total_cmp(sims[n_a + i], max_neg) > 0) max_neg = sims[n_a + i];
        return max_pos > max_neg ? scaled_fast_sigmoid(max_pos) : -scaled_fast_sigmoid(max_neg);
    }
    if (kind == 1) {
        float pos = 0.0f, neg = 0.0f;
        for (uint32_t i = 0; i < n_a; i++) pos += sims[i];
        for (uint32_t i = 0; i < n_b; i++) neg += sims[n_a + i];
        return pos - neg;
    }
    if (kind == 2) {
        int32_t rank = 0;
        for (uint32_t i = 0; i < n_b; i++) rank += f32_total_cmp(sims[1 + 2 * i], sims[2 + 2 * i]);
        return (float)rank + scaled_fast_sigmoid(sims[0]);
    }
    float sum = 0.0f;
    for (uint32_t i = 0; i < n_b; i++) {
        const float difference = sims[2 * i] - sims[2 * i + 1] - 1.1920929e-07f;    /* ScoreType::EPSILON */
        sum += fast_sigmoid(fminf(difference, 0.0f));
    }
    return sum;
}

/* FeedbackQuery::score_by (lib/segment/src/vector_storage/query/feedback_query.rs:198-226):
 *   let mut score = coefficients.a.0 * similarity(target);
 *   for pair in context_pairs { let delta = similarity(positive) - similarity(negative); score += partial_computation.0 * delta; } */
float qo_custom_feedback(uint32_t n_pairs, const float *sims, const float *coefs) {
    float score = coefs[0] * sims[0];
    for (uint32_t i = 0; i < n_pairs; i++) {
        const float delta = sims[1 + 2 * i] - sims[2 + 2 * i];
        score += coefs[1 + i] * delta;
    }
    return score;
}

/* ------------------------------------------------------------------------------------------
 * synthetic data: counter-based, integer-only (Irwin-Hall of four 16-bit uniforms), so the
 * device generator (qdrant_amd/csrc/synth.hip) reproduces it bit-for-bit without libm.
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
float qo_synth_value(uint64_t seed, uint64_t row, uint32_t col, uint32_t dim) {
    uint64_t h = splitmix64(splitmix64(seed) ^ (row * (uint64_t)dim + col));
    int32_t s = (int32_t)(h & 0xFFFF) + (int32_t)((h >> 16) & 0xFFFF) + (int32_t)((h >> 32) & 0xFFFF) + (int32_t)(h >> 48);
    return (float)(s - 131070) * (1.0f / 37837.0f);
}
void qo_synth_fill_f32(uint64_t seed, uint64_t row0, uint64_t n, uint32_t dim, float *out) {
    for (uint64_t r = 0; r < n; r++)
        for (uint32_t c = 0; c < dim; c++) out[r * dim + c] = qo_synth_value(seed, row0 + r, c, dim);
}
/* twin of qmx_synth_fill_latent_f32 (qdrant_amd/csrc/preprocess.hip synth_latent_kernel): one fmaf chain over the latent coordinates */
void qo_synth_fill_latent_f32(uint64_t seed, uint64_t row0, uint64_t n, uint32_t dim, uint32_t K, float noise, float *out) {
    float *W = (float *)malloc((size_t)K * dim * sizeof(float));
    float *z = (float *)malloc((size_t)K * sizeof(float));
    qo_synth_fill_f32(seed ^ 0x57ull, 0, K, dim, W);
    for (uint64_t r = 0; r < n; r++) {
        for (uint32_t k = 0; k < K; k++) z[k] = qo_synth_value(seed, row0 + r, k, K);
        float *o = out + r * dim;
        for (uint32_t c = 0; c < dim; c++) o[c] = 0.0f;
        for (uint32_t k = 0; k < K; k++) {
            const float zk = z[k];
            const float *w = W + (size_t)k * dim;
            for (uint32_t c = 0; c < dim; c++) o[c] = fmaf(zk, w[c], o[c]);
        }
        if (noise != 0.0f)
            for (uint32_t c = 0; c < dim; c++) o[c] = fmaf(noise, qo_synth_value(seed ^ 0xE5ull, row0 + r, c, dim), o[c]);
    }
    free(W);
    free(z);
}

/* ------------------------------------------------------------------------------------------
 * BQ: EncodedVectorsBin<u128> (lib/quantization/src/encoded_vectors_binary.rs), Encoding::OneBit,
 * QueryEncoding::SameAsStorage.  Test infrastructure only (see the header of this file).
 * ------------------------------------------------------------------------------------------ */
size_t qo_bq_row_bytes(uint32_t dim) {
    /* get_quantized_vector_size_from_params::<u128>(dim, OneBit) :829-840: get_storage_size(max(dim, 1)) * 16,
     * get_storage_size :412-419 = ceil(size / 128) */
    size_t size = dim > 1 ? dim : 1;
    size_t words = size / 128;
    if (size % 128 != 0) words += 1;
    return words * 16;
}

void qo_bq_encode_row(uint32_t dim, const float *v, uint8_t *out) {
    /* encode_vector :535-556 zero-fills, encode_one_bit_vector :558-568 sets bit (i % 128) of word (i / 128) when v > 0;
     * a little-endian u128 keeps bit b in byte b / 8, bit b % 8 */
    memset(out, 0, qo_bq_row_bytes(dim));
    for (uint32_t i = 0; i < dim; ++i) {
        if (v[i] > 0.0f) {
            const uint32_t word = i / 128, bit = i % 128;
            out[(size_t)word * 16 + bit / 8] |= (uint8_t)(1u << (bit % 8));
        }
    }
}

uint32_t qo_bq_xor_popcnt(const uint8_t *q, const uint8_t *v, uint32_t n_u128) {
    /* impl_xor_popcnt_sse_uint128 (cpp/sse.c:54-75): two u64 popcounts per word, summed in an i64 */
    int64_t result = 0;
    for (uint32_t w = 0; w < n_u128; ++w) {
        for (int h = 0; h < 2; ++h) {
            uint64_t a, b;
            memcpy(&a, q + (size_t)w * 16 + h * 8, 8);
            memcpy(&b, v + (size_t)w * 16 + h * 8, 8);
            result += __builtin_popcountll(a ^ b);
        }
    }
    return (uint32_t)result;
}

float qo_bq_score(int distance, int invert, uint32_t dim, const uint8_t *q, const uint8_t *v) {
    /* calculate_metric :766-810, query_bits_count == 1 */
    const float xor_product = (float)qo_bq_xor_popcnt(q, v, (uint32_t)(qo_bq_row_bytes(dim) / 16));
    const float fdim = (float)dim;
    const float zeros_count = fdim - xor_product;
    const int dot_like = distance == QO_DOT || distance == QO_COSINE;
    if (dot_like) return invert ? xor_product - zeros_count : zeros_count - xor_product;
    return invert ? zeros_count - xor_product : xor_product - zeros_count;
}

/* ------------------------------------------------------------------------------------------
 * find_quantile_interval (lib/quantization/src/quantile.rs:35-84) on a given sample.  Test infrastructure only.
 * ------------------------------------------------------------------------------------------ */
static int cmp_f32_partial(const void *a, const void *b) {   /* partial_cmp(..).unwrap_or(Equal) */
    const float x = *(const float *)a, y = *(const float *)b;
    return x < y ? -1 : x > y ? 1 : 0;
}
int qo_sq_quantile_interval(const float *sample, size_t n_sample, uint32_t dim, size_t count, float quantile, float *min_out, float *max_out) {
    if (count < 127 || quantile >= 1.0f) return 0;                                  /* :42-44 */
    const size_t len = n_sample * dim;                                              /* data_slice: the selected vectors, flattened */
    if (len < 4) return 0;                                                          /* :54-56 */
    size_t cut_index = (size_t)((float)n_sample * (1.0f - quantile) / 2.0f);       /* :58-62: selected_vectors_count as f32 * (1.0 - quantile) / 2.0 */
    if ((len - 1) / 2 < cut_index) cut_index = (len - 1) / 2;
    if (cut_index < 1) cut_index = 1;
    float *v = (float *)malloc(len * sizeof(float));
    memcpy(v, sample, len * sizeof(float));
    qsort(v, len, sizeof(float), cmp_f32_partial);
    /* select_nth_unstable(len - cut_index) -> left part = sorted [0, len - cut_index); select_nth_unstable(cut_index) on it ->
     * right part = sorted (cut_index, len - cut_index)                                                         :63-68 */
    const size_t lo = cut_index + 1, hi = len - cut_index;   /* [lo, hi) */
    if (hi < lo || hi - lo < 2) { free(v); return 0; }                              /* :70-72 */
    float mn = 3.4028235e38f, mx = -3.4028235e38f;                                  /* find_min_max_from_iter :19-33 */
    for (size_t i = lo; i < hi; ++i) { if (v[i] < mn) mn = v[i]; if (v[i] > mx) mx = v[i]; }
    free(v);
    *min_out = mn; *max_out = mx;
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * BQ, Encoding::TwoBits / OneAndHalfBits (encoded_vectors_binary.rs:570-672).  Test infrastructure only.
 * ------------------------------------------------------------------------------------------ */
size_t qo_bq_row_bytes_ex(uint32_t dim, int encoding) {
    /* extended_dim :833-838: dim, dim * 2, (dim * 3).div_ceil(2); get_storage_size(max(ext, 1)) * 16 */
    size_t ext = encoding == 0 ? dim : encoding == 1 ? (size_t)dim * 2 : ((size_t)dim * 3 + 1) / 2;
    if (ext < 1) ext = 1;
    size_t words = ext / 128;
    if (ext % 128 != 0) words += 1;
    return words * 16;
}
/* encode_two_bits_value :626-672 */
static void bq_two_bits_value(float value, const float *mean, const float *stddev, uint32_t i, int *b1, int *b2) {
    if (!mean || !stddev) { *b1 = *b2 = value > 0.0f; return; }       /* element_stats = None */
    const float sd = stddev[i];
    if (sd < 1.1920929e-07f) { *b1 = value > 0.0f; *b2 = 0; return; }  /* sd < f32::EPSILON */
    const float v_z = (value - mean[i]) / sd;
    const float SIGMAS = 2.0f / 3.0f;
    if (v_z <= -SIGMAS) { *b1 = 0; *b2 = 0; }
    else if (v_z < SIGMAS) { *b1 = 1; *b2 = 0; }
    else { *b1 = 1; *b2 = 1; }
}
static void bq_set_bit(uint8_t *out, size_t j) {   /* encoded_vector[j / 128] |= one << (j % 128), little-endian u128 */
    out[(j / 128) * 16 + (j % 128) / 8] |= (uint8_t)(1u << (j % 8));
}
/* VectorStatsBuilder::add for every row, then build (vector_stats.rs:48-102) */
void qo_vector_stats(const float *rows, uint64_t n, uint32_t dim, float *min, float *max, float *mean, float *stddev) {
    double *means = (double *)calloc(dim ? dim : 1, sizeof(double)), *m2 = (double *)calloc(dim ? dim : 1, sizeof(double));
    for (uint32_t d = 0; d < dim; d++) { min[d] = 3.40282347e+38f; max[d] = -3.40282347e+38f; }   /* f32::MAX, f32::MIN */
    uint64_t count = 0;
    for (uint64_t r = 0; r < n; r++) {
        count += 1;
        const double count_f64 = (double)count;
        const float *v = rows + r * dim;
        for (uint32_t d = 0; d < dim; d++) {
            const double value = (double)v[d];
            const float value_f32 = (float)value;
            if (value_f32 < min[d]) min[d] = value_f32;
            if (value_f32 > max[d]) max[d] = value_f32;
            const double delta = value - means[d];
            means[d] += delta / count_f64;
            m2[d] += delta * (value - means[d]);
        }
    }
    for (uint32_t d = 0; d < dim; d++) {
        stddev[d] = count > 1 ? (float)sqrt(m2[d] / (double)(count - 1)) : 0.0f;
        mean[d] = (float)means[d];
    }
    free(means);
    free(m2);
}

void qo_bq_encode_row_ex(uint32_t dim, int encoding, const float *mean, const float *stddev, const float *v, uint8_t *out) {
    memset(out, 0, qo_bq_row_bytes_ex(dim, encoding));
    for (uint32_t i = 0; i < dim; ++i) {
        if (encoding == 0) {                                   /* encode_one_bit_vector :558-568 */
            if (v[i] > 0.0f) bq_set_bit(out, i);
            continue;
        }
        int b1, b2;
        bq_two_bits_value(v[i], mean, stddev, i, &b1, &b2);
        if (b1) bq_set_bit(out, i);
        if (b2) bq_set_bit(out, encoding == 1 ? (size_t)dim + i : (size_t)dim + i / 2);   /* :570-624 */
    }
}
/* ---- QueryEncoding::Scalar4bits / Scalar8bits (encoded_vectors_binary.rs:49-54) ----
 * encode_scalar_query_vector (:692-719) + _encode_scalar_query_vector (:721-756): bytes of the encoded query =
 * get_storage_size(ext_len.max(1)) * bits u128 words; returns that byte count (call with out == NULL to size). */
size_t qo_bq_encode_scalar_query(uint32_t dim, int encoding, uint32_t bits, const float *query, uint8_t *out) {
    const size_t ext = encoding == 1 ? (size_t)dim * 2 : encoding == 2 ? (size_t)dim + ((size_t)dim + 1) / 2 : (size_t)dim;
    const size_t len1 = ext ? ext : 1;
    const size_t n_words = (len1 / 128 + (len1 % 128 ? 1 : 0)) * bits;          /* u128::get_storage_size :412-419 */
    if (!out) return n_words * 16;
    float *x = (float *)malloc(sizeof(float) * len1);
    for (uint32_t i = 0; i < dim; i++) x[i] = query[i];
    if (encoding == 1) for (uint32_t i = 0; i < dim; i++) x[dim + i] = query[i];
    if (encoding == 2) {
        for (uint32_t k = 0; 2 * k < dim; k++) {                                    /* chunks(2): max of the pair, or the odd last */
            const float a = query[2 * k];
            x[dim + k] = 2 * k + 1 < dim ? fmaxf(a, query[2 * k + 1]) : a;
        }
    }
    memset(out, 0, n_words * 16);
    float max_abs = 0.0f;
    for (size_t i = 0; i < ext; i++) max_abs = fmaxf(max_abs, fabsf(x[i]));        /* fold(0.0, f32::max) */
    const float mn = -max_abs, mx = max_abs;
    const size_t ranges = ((size_t)1 << bits) - 1;
    const float delta = (mx - mn) / (float)ranges;
    for (size_t i = 0; i < ext; i++) {
        const size_t chunk = i / 128, shift = i % 128;
        const float shifted = x[i] - mn;
        const float delted = delta > 1.1920929e-07f ? shifted / delta : 0.0f;
        const float r = roundf(delted);
        const size_t rounded = !(r >= 0.0f) ? 0 : (r >= 1.8446744e19f ? (size_t)-1 : (size_t)r);   /* `as usize` saturates, NaN -> 0 */
        const size_t quantized = rounded % (ranges + 1);
        for (uint32_t b = 0; b < bits; b++)
            if ((quantized >> b) & 1) out[(bits * chunk + b) * 16 + shift / 8] |= (uint8_t)(1u << (shift % 8));
    }
    free(x);
    return n_words * 16;
}
/* xor_popcnt_scalar, the scalar loop (:403-409): vector = n_u128 words, query = n_u128 * bits words */
uint64_t qo_bq_xor_popcnt_scalar(const uint8_t *vector, const uint8_t *query, uint32_t n_u128, uint32_t bits) {
    uint64_t result = 0;
    for (uint32_t w = 0; w < n_u128; w++) {
        uint64_t v[2];
        memcpy(v, vector + (size_t)w * 16, 16);
        for (uint32_t b = 0; b < bits; b++) {
            uint64_t q[2];
            memcpy(q, query + ((size_t)w * bits + b) * 16, 16);
            result += (uint64_t)(__builtin_popcountll(v[0] ^ q[0]) + __builtin_popcountll(v[1] ^ q[1])) << b;
        }
    }
    return result;
}
/* calculate_metric with query_bits_count = bits (:783-810) */
float qo_bq_score_scalar(int distance, int invert, uint32_t dim, int encoding, uint32_t bits, const uint8_t *scalar_query, const uint8_t *v) {
    const uint64_t x = qo_bq_xor_popcnt_scalar(v, scalar_query, (uint32_t)(qo_bq_row_bytes_ex(dim, encoding) / 16), bits);
    const float xor_product = (float)x / (float)((1 << bits) - 1);
    const float fdim = (float)dim;
    const float zeros_count = fdim - xor_product;
    const int dot_like = distance == QO_DOT || distance == QO_COSINE;
    if (dot_like) return invert ? xor_product - zeros_count : zeros_count - xor_product;
    return invert ? zeros_count - xor_product : xor_product - zeros_count;
}
float qo_bq_score_ex(int distance, int invert, uint32_t dim, int encoding, const uint8_t *q, const uint8_t *v) {
    const float xor_product = (float)qo_bq_xor_popcnt(q, v, (uint32_t)(qo_bq_row_bytes_ex(dim, encoding) / 16));
    const float fdim = (float)dim;                             /* metadata.vector_parameters.dim: the original dimension */
    const float zeros_count = fdim - xor_product;
    const int dot_like = distance == QO_DOT || distance == QO_COSINE;
    if (dot_like) return invert ? xor_product - zeros_count : zeros_count - xor_product;
    return invert ? zeros_count - xor_product : xor_product - zeros_count;
}

/* ---- Colbert MaxSim over multi-dense vectors: score_max_similarity (lib/segment/src/vector_storage/query_scorer/mod.rs:70-97) ----
 * sims[a * stride + b] = TMetric::similarity(dense_a, dense_b) (computed by the leaves above); the two loops of the reference. */
float qo_max_similarity(const float *sims, uint32_t n_a, uint32_t n_b, uint64_t stride) {
    float sum = 0.0f;
    for (uint32_t a = 0; a < n_a; a++) {
        float max_sim = -INFINITY;
        for (uint32_t b = 0; b < n_b; b++) {
            const float sim = sims[(uint64_t)a * stride + b];
            if (sim > max_sim) max_sim = sim;
        }
        sum += max_sim;
    }
    return sum;
}


/* oracle/qdrant_oracle_tq.c — TEST INFRASTRUCTURE (see qdrant_oracle.h): CPU restatement of the reference's TurboQuant quantizer,
 * lib/quantization/src/turboquant/ (v1.19.0) as used by EncodedVectorsTQ (lib/quantization/src/encoded_vectors_tq.rs).
 *
 *   permutation.rs:9-36,108-115,151-153   ReversibleLcg (Knuth MMIX), Permutation::permute (Fisher-Yates replay), bounded_rand
 *   rotation.rs:4-10,32-77,97-131,222-233,264-280   PERMUTATION_SEEDS, HadamardRotation::{new, apply}, wht_and_gather_rounds,
 *                                          compute_chunk_sizes, wht_normalized_chunks; in_place_walsh_hadamard_transform :158-176
 *                                          (the SIMD variants of simd/hadamard.rs are bit-equal to it by the crate's own tests)
 *   lloyd_max.rs:3-31                      CENTROIDS_{1,2,4}BIT and their midpoint boundaries
 *   quantization.rs:160-209,211-296,299-318   preprocess_into, quantize_impl (TQMode::Normal), compute_centroid_norm
 *   encoding.rs:117-134,194-206,211-221,231-258   pack_vector, padded_dim, compute_l2_length, pack_extras_into (+ size_for :31-56)
 *   quantization.rs:496-567,569-620        precompute_query, score_precomputed (Dot / Cosine / L2)
 *   quantization.rs:395-445                score_symmetric (Dot / Cosine / L2)
 *   simd/query4bit/mod.rs (x86_64 constants :64-90, new :186-248, dotprod :257-261, dotprod_raw :300-346, score_4bit_internal_* :414-434)
 *   simd/query2bit/mod.rs, simd/query1bit/mod.rs   the same for 2 bits and for the bit-plane form of 1 bit (BITS = 8)
 *
 * Scope: TQMode::Normal and TQMode::Plus with GIVEN ErrorCorrection shift / scale (the reference fits them with P-square quantile estimators over
 * randomly sampled vectors, encoded_vectors_tq.rs:156-240, and persists them in the metadata: an input, like PQ centroids), TQBits 4 / 2 / 1.5 / 1, TQRotation Padded and Unpadded, distances Dot, Cosine, L2.  L1 (full dequantisation + inverse
 * rotation per score) is not restated.
 *
 * PARITY UNPINNED at the bit level: the reference holds tolerance tests only for this quantizer (tests/integration/test_tq.rs: error =
 * coef(bits) * signal_std) plus literals for the chunk decomposition (rotation.rs:283-315) and the codebooks; all of them are mirrored in
 * tests/test_oracle_tq.py.  The integer kernels are exact in any order, so the only order-sensitive parts are the f64 sums, restated in
 * the reference's sequential order. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "qdrant_oracle.h"

enum { TQ_BITS4 = 0, TQ_BITS2 = 1, TQ_BITS1_5 = 2, TQ_BITS1 = 3 };

static const float CENTROIDS_1BIT[2] = {-0.7978846f, 0.7978846f};
static const float CENTROIDS_2BIT[4] = {-1.510f, -0.4528f, 0.4528f, 1.510f};
static const float CENTROIDS_4BIT[16] = {-2.733f, -2.069f, -1.618f, -1.256f, -0.9424f, -0.6568f, -0.3881f, -0.1284f,
                                         0.1284f, 0.3881f, 0.6568f, 0.9424f, 1.256f, 1.618f, 2.069f, 2.733f};
/* x86_64 integer codebooks (simd/query{4,2}bit/mod.rs): c_u = c_signed + 128 */
static const uint8_t CODEBOOK_U8_4BIT[16] = {0, 31, 52, 69, 84, 97, 110, 122, 134, 146, 159, 172, 187, 204, 225, 255};
static const uint8_t CODEBOOK_U8_2BIT[4] = {0, 90, 166, 255};
#define TQ_CODEBOOK_OFFSET 128
#define TQ_QUERY_ABS_MAX 8127.0f
#define TQ_QUERY_HIGH_COEF 128

struct qo_tq {
    uint32_t dim, padded_dim, rot_dim;
    int bits, distance;
    uint32_t *maps[3];     /* forward_maps */
    /* TQ+ (TQMode::Plus): ErrorCorrection (quantization.rs:28-96), NULL without */
    float *shift, *scale;
    int16_t *d_prime_sq_i16;
    float weight_scale, mm_const;
};

static int bit_size(int bits) { return bits == TQ_BITS4 ? 4 : bits == TQ_BITS2 ? 2 : 1; }
static const float *centroids_of(int bits, int *n) {
    if (bits == TQ_BITS4) { *n = 16; return CENTROIDS_4BIT; }
    if (bits == TQ_BITS2) { *n = 4; return CENTROIDS_2BIT; }
    *n = 2;
    return CENTROIDS_1BIT;
}
static uint32_t next_multiple(uint32_t x, uint32_t m) { return (x + m - 1) / m * m; }
uint32_t qo_tq_padded_dim_for(uint32_t dim, int bits) {           /* encoding.rs:194-201 */
    switch (bits) {
        case TQ_BITS1: return next_multiple(dim, 8);
        case TQ_BITS1_5: return next_multiple(dim * 3 / 2, 8);
        case TQ_BITS2: return next_multiple(dim, 4);
        default: return next_multiple(dim, 2);
    }
}
static uint32_t extras_size_mode(int distance, int plus) { return (distance == QO_EUCLID ? 8u : 4u) + (plus ? 4u : 0u); }   /* size_for (encoding.rs:31-56) */

/* permutation.rs: forward map of Permutation::new_one_way(seed, count).permute(identity) */
void qo_tq_permutation_map(uint64_t seed, uint32_t count, uint32_t *map) {
    for (uint32_t i = 0; i < count; i++) map[i] = i;
    uint64_t state = seed;
    for (uint32_t i = count; i-- > 1;) {                  /* (1..count).rev().zip(rng) */
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        const uint32_t j = (uint32_t)((state >> 32) % ((uint64_t)i + 1));
        const uint32_t t = map[i]; map[i] = map[j]; map[j] = t;
    }
}

uint32_t qo_tq_chunk_sizes(uint32_t dim, uint32_t *out) {  /* compute_chunk_sizes: decreasing powers of two summing to dim */
    uint32_t n = 0, bits = dim;
    while (bits) {
        uint32_t highest = 1u << (31 - __builtin_clz(bits));
        bits ^= highest;
        out[n++] = highest;
    }
    return n;
}

void qo_tq_wht(double *x, uint32_t n) {                    /* in_place_walsh_hadamard_transform */
    for (uint32_t h = 1; h < n; h *= 2)
        for (uint32_t i = 0; i < n; i += h * 2)
            for (uint32_t j = i; j < i + h; j++) {
                const double a = x[j], b = x[j + h];
                x[j] = a + b;
                x[j + h] = a - b;
            }
}
static void wht_normalized_chunks(double *buf, uint32_t len) {
    uint32_t sizes[32];
    const uint32_t n = qo_tq_chunk_sizes(len, sizes);
    uint32_t off = 0;
    for (uint32_t c = 0; c < n; c++) {
        const double norm = 1.0 / sqrt((double)sizes[c]);
        qo_tq_wht(buf + off, sizes[c]);
        for (uint32_t i = 0; i < sizes[c]; i++) buf[off + i] *= norm;
        off += sizes[c];
    }
}

qo_tq *qo_tq_new(uint32_t dim, int bits, int distance, int rotation_unpadded) {
    qo_tq *t = (qo_tq *)calloc(1, sizeof(qo_tq));
    static const uint64_t SEEDS[3] = {654605292835415893ull, 8636605637963351413ull, 1775280196666917949ull};
    t->dim = dim; t->bits = bits; t->distance = distance;
    t->padded_dim = qo_tq_padded_dim_for(dim, bits);
    t->rot_dim = rotation_unpadded ? dim : t->padded_dim;
    for (int p = 0; p < 3; p++) {
        t->maps[p] = (uint32_t *)malloc(sizeof(uint32_t) * (t->rot_dim ? t->rot_dim : 1));
        qo_tq_permutation_map(SEEDS[p], t->rot_dim, t->maps[p]);
    }
    return t;
}
/* ErrorCorrection::new (quantization.rs:49-96): shift / scale [padded_dim] as persisted in the metadata (encoded_vectors_tq.rs:93-96) */
qo_tq *qo_tq_new_plus(uint32_t dim, int bits, int distance, int rotation_unpadded, const float *shift, const float *scale) {
    qo_tq *t = qo_tq_new(dim, bits, distance, rotation_unpadded);
    const uint32_t pd = t->padded_dim;
    t->shift = (float *)malloc(sizeof(float) * pd);
    t->scale = (float *)malloc(sizeof(float) * pd);
    t->d_prime_sq_i16 = (int16_t *)malloc(sizeof(int16_t) * pd);
    memcpy(t->shift, shift, sizeof(float) * pd);
    memcpy(t->scale, scale, sizeof(float) * pd);
    float mm = 0.0f;                                             /* shift.iter().map(|&s| s * s).sum() (f32, in order) */
    for (uint32_t i = 0; i < pd; i++) mm += shift[i] * shift[i];
    t->mm_const = mm;
    float *dps = (float *)malloc(sizeof(float) * pd);
    float max_dps = 0.0f;
    for (uint32_t i = 0; i < pd; i++) {
        dps[i] = fabsf(scale[i]) > 1.1920929e-7f ? 1.0f / (scale[i] * scale[i]) : 0.0f;      /* (s * s).recip() */
        if (dps[i] > max_dps) max_dps = dps[i];
    }
    const float QUANT_CAP = 32766.0f;                           /* i16::MAX - 1 */
    t->weight_scale = max_dps > 1.1920929e-7f ? QUANT_CAP / max_dps : 1.0f;
    for (uint32_t i = 0; i < pd; i++) {
        float v = roundf(dps[i] * t->weight_scale);
        if (v < 0.0f) v = 0.0f;
        if (v > QUANT_CAP) v = QUANT_CAP;
        t->d_prime_sq_i16[i] = (int16_t)v;
    }
    free(dps);
    return t;
}
void qo_tq_free(qo_tq *t) {
    if (!t) return;
    for (int p = 0; p < 3; p++) free(t->maps[p]);
    free(t->shift); free(t->scale); free(t->d_prime_sq_i16);
    free(t);
}
uint32_t qo_tq_padded_dim(const qo_tq *t) { return t->padded_dim; }
uint32_t qo_tq_quantized_size(const qo_tq *t) { return t->padded_dim * bit_size(t->bits) / 8 + extras_size_mode(t->distance, t->shift != NULL); }

/* HadamardRotation::apply on buf[..rot_dim] */
void qo_tq_rotate(const qo_tq *t, double *x) {
    const uint32_t n = t->rot_dim;
    if (n == 0) return;
    double *scratch = (double *)malloc(sizeof(double) * n);
    wht_normalized_chunks(x, n);
    double *src = x, *dst = scratch;
    for (int p = 0; p < 3; p++) {
        for (uint32_t k = 0; k < n; k++) dst[k] = src[t->maps[p][k]];
        wht_normalized_chunks(dst, n);
        double *s = src; src = dst; dst = s;
    }
    if (src != x) memcpy(x, src, sizeof(double) * n);
    free(scratch);
}

static uint32_t centroid_index(const float *centroids, int n, double val) {   /* boundaries.partition_point(|&b| (val as f32) > b) */
    const float v = (float)val;
    uint32_t idx = 0;
    for (int i = 0; i + 1 < n; i++) {
        const float b = (centroids[i] + centroids[i + 1]) / 2.0f;
        if (v > b) idx = (uint32_t)i + 1; else break;
    }
    return idx;
}

/* ---- TQ+ parameter fit: EncodedVectorsTQ::encode's first pass (encoded_vectors_tq.rs:156-234) over
 * find_quantile_interval_per_coordinate_with_preprocess (quantile.rs:130-281) and the extended P-square estimator
 * (p_square.rs, Jain & Chlamtac with P2_MARKERS = 7 markers, x86_64 AVX2 + FMA dispatch restated: the desired positions of markers
 * 0..3 are one fused multiply-add, markers 4..6 `1.0 + p * (count - 1)` in two roundings; find_marker counts heights[1..N-1] < x).
 * Which vectors are sampled (Permutor, a third-party permutation iterator; bits.sample_size() of them, ascending index order) is an
 * input: `sample` holds them in that order.  Parity unpinned: the reference's tests of this code are statistical. ---- */
#define QO_P2_N 7
typedef struct { int count; double obs[QO_P2_N]; double q, h[QO_P2_N], n[QO_P2_N], nd[QO_P2_N], tp[QO_P2_N]; } qo_p2;
static void p2_init(qo_p2 *e, double q) { memset(e, 0, sizeof(*e)); e->q = q; }
static int cmp_f64(const void *a, const void *b) { const double x = *(const double *)a, y = *(const double *)b; return x < y ? -1 : x > y ? 1 : 0; }
static void p2_grid(double q, double *p) {     /* generate_grid_probabilities (p_square.rs:171-214) */
    const int extra = (QO_P2_N - 5) / 2;
    p[0] = 0.0;
    p[1] = q * 0.5;
    for (int i = 0; i < extra; i++) p[i + 2] = q * (0.7 + 0.3 * (double)(i + 1) / ((double)extra + 2.0));
    p[QO_P2_N / 2] = q;
    for (int i = 0; i < extra; i++) p[QO_P2_N / 2 + 1 + i] = 1.0 + (q - 1.0) * (0.7 + 0.3 * (double)(extra - i) / ((double)extra + 2.0));
    p[QO_P2_N - 2] = 1.0 + (q - 1.0) * 0.5;
    p[QO_P2_N - 1] = 1.0;
}
static void p2_adjust_step(qo_p2 *e, int i, double dsign) {     /* adjust_step (:456-488) */
    const double prev_h = e->h[i - 1], next_h = e->h[i + 1], prev_n = e->n[i - 1], next_n = e->n[i + 1], cur_h = e->h[i], cur_n = e->n[i];
    const double denom = next_n - prev_n;
    double h_par = cur_h;
    if (denom != 0.0) {
        const double a = (cur_n - prev_n + dsign) / (next_n - cur_n) * (next_h - cur_h);
        const double b = (next_n - cur_n - dsign) / (cur_n - prev_n) * (cur_h - prev_h);
        h_par = cur_h + (a + b) * dsign / denom;
    }
    if (h_par > prev_h && h_par < next_h && isfinite(h_par)) e->h[i] = h_par;
    else if (dsign > 0.0) e->h[i] = cur_h + (next_h - cur_h) / (next_n - cur_n);
    else e->h[i] = cur_h + (prev_h - cur_h) / (prev_n - cur_n);
    e->n[i] += dsign;
}
static void p2_push(qo_p2 *e, double x) {      /* P2Quantile::push (:40-58) + P2QuantileImpl::push (:123-160) */
    if (isnan(x) || !isfinite(x)) return;
    if (e->count < QO_P2_N) {
        e->obs[e->count++] = x;
        if (e->count == QO_P2_N) {             /* new_from_linear (:91-116) */
            double buf[QO_P2_N];
            memcpy(buf, e->obs, sizeof(buf));
            qsort(buf, QO_P2_N, sizeof(double), cmp_f64);
            p2_grid(e->q, e->tp);
            for (int i = 0; i < QO_P2_N; i++) {
                e->h[i] = buf[i];
                e->n[i] = (double)(i + 1);
                e->nd[i] = 1.0 + e->tp[i] * (double)(QO_P2_N - 1);
            }
        }
        return;
    }
    e->count += 1;
    int k;
    if (x < e->h[0]) { e->h[0] = x; k = 0; }
    else if (x > e->h[QO_P2_N - 1]) { e->h[QO_P2_N - 1] = x; k = QO_P2_N - 1; }
    else { k = 0; for (int i = 1; i < QO_P2_N; i++) if (e->h[i] < x) k++; }     /* find_marker_avx2 (:369-397) */
    for (int i = k + 1; i < QO_P2_N; i++) e->n[i] += 1.0;
    const double cm1 = (double)(e->count - 1);
    for (int i = 0; i < QO_P2_N; i++)          /* update_desired_avx2 (:401-432): 4-lane fmadd, scalar tail */
        e->nd[i] = i < (QO_P2_N / 4) * 4 ? fma(e->tp[i], cm1, 1.0) : 1.0 + e->tp[i] * cm1;
    for (int i = 1; i < QO_P2_N - 1; i++) {    /* adjust_marker (:438-454) */
        for (;;) {
            const double di = e->nd[i] - e->n[i];
            if (di >= 1.0 && (e->n[i + 1] - e->n[i]) > 1.0) p2_adjust_step(e, i, 1.0);
            else if (di <= -1.0 && (e->n[i - 1] - e->n[i]) < -1.0) p2_adjust_step(e, i, -1.0);
            else break;
        }
    }
}
static double p2_estimate(qo_p2 *e) {          /* estimate (:61-66,118-121) / estimate_quantile_from_slice (:502-521) */
    if (e->count >= QO_P2_N) return e->h[QO_P2_N / 2];
    if (e->count == 0) return 0.0;
    if (e->count == 1) return e->obs[0];
    double buf[QO_P2_N];
    memcpy(buf, e->obs, sizeof(double) * (size_t)e->count);
    qsort(buf, (size_t)e->count, sizeof(double), cmp_f64);
    const double k = e->q * ((double)e->count - 1.0);
    const size_t lo = (size_t)floor(k), hi = (size_t)ceil(k);
    if (lo == hi) return buf[lo];
    return buf[lo] + (k - (double)lo) * (buf[hi] - buf[lo]);
}
/* one estimator over a stream (the reference's own tests drive it this way, p_square.rs:631-812); grid = its 7 target probabilities */
double qo_p2_quantile(double q, const double *values, uint64_t n, double *grid) {
    qo_p2 e;
    p2_init(&e, q);
    for (uint64_t i = 0; i < n; i++) p2_push(&e, values[i]);
    if (grid) p2_grid(q, grid);
    return p2_estimate(&e);
}
/* turboquant/math.rs:3-15 */
static double std_normal_cdf(double x) {
    const double y = x / 1.4142135623730951;   /* std::f64::consts::SQRT_2 */
    const double a = fabs(y);
    const double t = 1.0 / (1.0 + 0.3275911 * a);
    const double poly = t * (0.254829592 + t * (-0.284496736 + t * (1.421413741 + t * (-1.453152027 + t * 1.061405429))));
    const double r = 1.0 - poly * exp(-a * a);
    return 0.5 * (1.0 + (y >= 0.0 ? r : -r));
}
/* the two quantiles the estimators track for `bits` (encoded_vectors_tq.rs:172-184 + quantile.rs:155-156) and the outermost centroid */
void qo_tq_plus_quantiles(int bits, double *min_q, double *max_q, float *c_outer_out) {
    int nc;
    const float *centroids = centroids_of(bits, &nc);
    float c_outer = 0.0f;
    for (int i = 0; i < nc; i++) { const float a = fabsf(centroids[i]); c_outer = c_outer > a ? c_outer : a; }   /* fold(0.0, |acc, c| acc.max(c.abs())) */
    const double p_outer = std_normal_cdf((double)c_outer);
    float qp = (float)(2.0 * p_outer - 1.0);
    qp = qp < 0.0f ? 0.0f : qp > 0.99999f ? 0.99999f : qp;
    *min_q = (1.0 - (double)qp) / 2.0;
    *max_q = 1.0 - *min_q;
    *c_outer_out = c_outer;
}
/* TurboQuantizer::preprocess_into (quantization.rs:169-207): pad, rotate, rescale to norm sqrt(padded_dim); buf[padded_dim] */
void qo_tq_preprocess(const qo_tq *t, const float *vec, double *buf) {
    const uint32_t pd = t->padded_dim;
    for (uint32_t i = 0; i < pd; i++) buf[i] = i < t->dim ? (double)vec[i] : 0.0;
    qo_tq_rotate(t, buf);
    float l2_length = 1.0f;
    if (t->distance != QO_COSINE) {
        double s = 0.0;
        for (uint32_t i = 0; i < pd; i++) s += buf[i] * buf[i];
        l2_length = (float)sqrt(s);
    }
    const double length = (double)l2_length;
    if (length > 0.0) {
        const double length_scale = sqrt((double)pd) / length;
        for (uint32_t i = 0; i < pd; i++) buf[i] *= length_scale;
    }
}
/* shift / scale [padded_dim] from `n_sample` sampled vectors in their iteration order; `t` = the TQMode::Normal pre-quantizer */
void qo_tq_plus_fit(const qo_tq *t, const float *sample, uint32_t n_sample, float *shift, float *scale) {
    const uint32_t pd = t->padded_dim;
    double min_q, max_q;
    float c_outer;
    qo_tq_plus_quantiles(t->bits, &min_q, &max_q, &c_outer);
    qo_p2 *est = (qo_p2 *)malloc(sizeof(qo_p2) * 2 * (size_t)pd);
    for (uint32_t d = 0; d < pd; d++) { p2_init(&est[2 * d], min_q); p2_init(&est[2 * d + 1], max_q); }
    double *buf = (double *)malloc(sizeof(double) * (pd ? pd : 1));
    for (uint32_t v = 0; v < n_sample; v++) {
        qo_tq_preprocess(t, sample + (size_t)v * t->dim, buf);
        for (uint32_t d = 0; d < pd; d++) { p2_push(&est[2 * d], buf[d]); p2_push(&est[2 * d + 1], buf[d]); }
    }
    for (uint32_t d = 0; d < pd; d++) {
        float q_lo = 0.0f, q_hi = 0.0f;
        if (n_sample) { q_lo = (float)p2_estimate(&est[2 * d]); q_hi = (float)p2_estimate(&est[2 * d + 1]); }   /* count == 0: zero pairs (:151-153) */
        shift[d] = -(q_lo + q_hi) / 2.0f;
        const float denom = q_hi - q_lo;
        scale[d] = denom > 1e-3f ? (2.0f * c_outer) / denom : 1.0f;     /* MIN_QUANTILE_WIDTH */
    }
    free(buf);
    free(est);
}

/* TurboQuantizer::quantize (TQMode::Normal): out = [codes][scaling_factor f32][l2_length f32 (L2 only)] */
void qo_tq_quantize(const qo_tq *t, const float *vec, uint8_t *out) {
    const uint32_t pd = t->padded_dim;
    double *buf = (double *)calloc(pd ? pd : 1, sizeof(double));
    for (uint32_t i = 0; i < t->dim; i++) buf[i] = (double)vec[i];
    qo_tq_rotate(t, buf);
    int has_l2 = t->distance != QO_COSINE;
    float l2_length = 1.0f;
    if (has_l2) {
        double s = 0.0;
        for (uint32_t i = 0; i < pd; i++) s += buf[i] * buf[i];
        l2_length = (float)sqrt(s);
    }
    const double length = (double)l2_length;
    if (length > 0.0) {
        const double length_scale = sqrt((double)pd) / length;
        for (uint32_t i = 0; i < pd; i++) buf[i] *= length_scale;
    }
    /* TQ+: xm = <X, M> on the rescaled vector, then the per-coordinate shift + scale (skipped for an all-zero vector) :231-247 */
    float xm = 0.0f;
    int ec_applied = 0;
    if (t->shift) {
        double l2sq = 0.0;
        for (uint32_t i = 0; i < pd; i++) l2sq += buf[i] * buf[i];
        if (!(l2sq < 1e-12)) {
            double x = 0.0;
            for (uint32_t i = 0; i < pd; i++) x += buf[i] * (double)(-t->shift[i]);
            for (uint32_t i = 0; i < pd; i++) buf[i] = (buf[i] + (double)t->shift[i]) * (double)t->scale[i];
            xm = (float)x;
            ec_applied = 1;
        }
    }
    (void)ec_applied;
    int nc;
    const float *centroids = centroids_of(t->bits, &nc);
    /* centroid norm (Dot / L2: always; Cosine: sqrt(padded_dim) for a zero vector) */
    float centroid_norm;
    {
        int degenerate = 0;
        if (t->distance == QO_COSINE) {
            double s = 0.0;
            for (uint32_t i = 0; i < pd; i++) s += buf[i] * buf[i];
            degenerate = s < 1e-12;
        }
        if (degenerate) centroid_norm = sqrtf((float)pd);
        else {
            double sq = 0.0;
            for (uint32_t i = 0; i < pd; i++) {
                double c = (double)centroids[centroid_index(centroids, nc, buf[i])];
                if (t->shift) c = c / (double)t->scale[i] - (double)t->shift[i];     /* compute_centroid_norm reverts the EC :311-314 */
                sq += c * c;
            }
            centroid_norm = (float)sqrt(sq);
        }
    }
    const uint32_t bs = (uint32_t)bit_size(t->bits), code_bytes = pd * bs / 8;
    memset(out, 0, code_bytes);
    for (uint32_t i = 0; i < pd; i++) {                       /* BitWriter, LSB first */
        const uint32_t idx = centroid_index(centroids, nc, buf[i]), bit = i * bs;
        out[bit / 8] |= (uint8_t)(idx << (bit % 8));
    }
    /* pack_extras_into (encoding.rs:218-247): l2 / centroid_norm; L1 stores the bare l2 length (no centroid norm, quantization.rs:275) */
    const float scaling_factor = t->distance == QO_MANHATTAN ? l2_length : (has_l2 ? l2_length : 1.0f) / centroid_norm;
    memcpy(out + code_bytes, &scaling_factor, 4);
    if (t->distance == QO_EUCLID) memcpy(out + code_bytes + 4, &l2_length, 4);
    if (t->shift) memcpy(out + code_bytes + (t->distance == QO_EUCLID ? 8 : 4), &xm, 4);      /* the trailing f32 of the extras */
    free(buf);
}

struct qo_tq_query {
    int32_t *q;            /* q_signed per padded dim */
    float postprocess_scale, l2_norm, ec_correction;
    int64_t sum_q;
    float *rotated_f32;
    float *raw;            /* DistanceType::L1 keeps the query as given (quantization.rs:532-535), NULL otherwise */
};

qo_tq_query *qo_tq_precompute_query(const qo_tq *t, const float *query) {
    const uint32_t pd = t->padded_dim;
    qo_tq_query *e = (qo_tq_query *)calloc(1, sizeof(qo_tq_query));
    double *rot = (double *)calloc(pd ? pd : 1, sizeof(double));
    for (uint32_t i = 0; i < t->dim; i++) rot[i] = (double)query[i];
    qo_tq_rotate(t, rot);
    e->l2_norm = 1.0f;
    if (t->distance != QO_COSINE) {
        double s = 0.0;
        for (uint32_t i = 0; i < pd; i++) s += rot[i] * rot[i];
        e->l2_norm = (float)sqrt(s);
    }
    if (t->shift) {        /* qm = <Q, M>, then Q .* D' (:526-540) */
        double qm = 0.0;
        for (uint32_t i = 0; i < pd; i++) qm += rot[i] * (double)(-t->shift[i]);
        for (uint32_t i = 0; i < pd; i++) rot[i] /= (double)t->scale[i];
        e->ec_correction = (float)qm;
    }
    e->rotated_f32 = (float *)malloc(sizeof(float) * (pd ? pd : 1));
    e->q = (int32_t *)malloc(sizeof(int32_t) * (pd ? pd : 1));
    float q_abs_max = 0.0f;
    for (uint32_t i = 0; i < pd; i++) {
        e->rotated_f32[i] = (float)rot[i];
        const float a = fabsf(e->rotated_f32[i]);
        if (a > q_abs_max) q_abs_max = a;                      /* fold(0.0, f32::max) */
    }
    if (!(q_abs_max > 1.1920929e-7f)) q_abs_max = 1.1920929e-7f;   /* .max(f32::EPSILON) */
    const int one_bit = t->bits == TQ_BITS1 || t->bits == TQ_BITS1_5;
    /* (1 << (BITS - 1)) - 1 with BITS = 8; TQ+ over 1-bit storage widens the query to Query1bitSimd<16> (:557-563) */
    const float abs_max_int = one_bit ? (t->shift ? 32767.0f : 127.0f) : TQ_QUERY_ABS_MAX;
    const float q_scale = abs_max_int / q_abs_max;
    for (uint32_t i = 0; i < pd; i++) {
        float v = roundf(e->rotated_f32[i] * q_scale);
        if (v > abs_max_int) v = abs_max_int;
        if (v < -abs_max_int) v = -abs_max_int;
        e->q[i] = (int32_t)v;
        e->sum_q += e->q[i];
    }
    if (one_bit) e->postprocess_scale = 0.7978846f / q_scale;                         /* CENTROID_ABS / q_scale */
    else {
        const float codebook_scale = 128.0f / (t->bits == TQ_BITS4 ? 2.733f : 1.510f);
        e->postprocess_scale = 1.0f / (q_scale * codebook_scale);
    }
    if (t->distance == QO_MANHATTAN) {
        e->raw = (float *)malloc(sizeof(float) * (t->dim ? t->dim : 1));
        memcpy(e->raw, query, sizeof(float) * t->dim);
    }
    free(rot);
    return e;
}
void qo_tq_query_free(qo_tq_query *e) {
    if (!e) return;
    free(e->q); free(e->rotated_f32); free(e->raw); free(e);
}
/* the encoded query as the device holds it: q_signed [padded_dim], postprocess_scale, l2_norm, sum of q_signed */
float qo_tq_query_ec_correction(const qo_tq_query *e) { return e->ec_correction; }
void qo_tq_query_export(const qo_tq *t, const qo_tq_query *e, int32_t *q_out, float *postprocess_scale, float *l2_norm, int64_t *sum_q) {
    if (q_out) memcpy(q_out, e->q, sizeof(int32_t) * t->padded_dim);
    if (postprocess_scale) *postprocess_scale = e->postprocess_scale;
    if (l2_norm) *l2_norm = e->l2_norm;
    if (sum_q) *sum_q = e->sum_q;
}

static uint32_t code_at(const uint8_t *codes, uint32_t i, uint32_t bs) { return (codes[i * bs / 8] >> (i * bs % 8)) & ((1u << bs) - 1u); }

/* Query{N}bitSimd::dotprod: the float "raw_dot" */
static float tq_raw_dot(const qo_tq *t, const qo_tq_query *e, const uint8_t *codes) {
    const uint32_t pd = t->padded_dim;
    if (t->bits == TQ_BITS1 || t->bits == TQ_BITS1_5) {
        int64_t v_dot_q = 0;                                   /* sum over set bits of q (the two's-complement bit planes add up to q) */
        for (uint32_t i = 0; i < pd; i++) if (code_at(codes, i, 1)) v_dot_q += e->q[i];
        const int64_t signed_dot = 2 * v_dot_q - e->sum_q;
        return e->postprocess_scale * (float)signed_dot;
    }
    const uint32_t bs = (uint32_t)bit_size(t->bits);
    const uint8_t *book = t->bits == TQ_BITS4 ? CODEBOOK_U8_4BIT : CODEBOOK_U8_2BIT;
    int64_t dot_raw = 0;                                       /* acc_low + 128 acc_high = sum q_signed * c_u */
    for (uint32_t i = 0; i < pd; i++) dot_raw += (int64_t)e->q[i] * (int64_t)book[code_at(codes, i, bs)];
    const int64_t bias = (int64_t)TQ_CODEBOOK_OFFSET * e->sum_q;
    return e->postprocess_scale * (float)(dot_raw - bias);
}

/* HadamardRotation::apply_inverse on x[..rot_dim] (rotation.rs:76-79): the same WHT / gather rounds over the backward maps, last map first;
 * backward_maps[p][forward_maps[p][k]] = k (rotation.rs:47-53) */
void qo_tq_rotate_inverse(const qo_tq *t, double *x) {
    const uint32_t n = t->rot_dim;
    if (n == 0) return;
    double *scratch = (double *)malloc(sizeof(double) * n);
    uint32_t *inv = (uint32_t *)malloc(sizeof(uint32_t) * n);
    wht_normalized_chunks(x, n);
    double *src = x, *dst = scratch;
    for (int p = 2; p >= 0; p--) {
        for (uint32_t k = 0; k < n; k++) inv[t->maps[p][k]] = k;
        for (uint32_t k = 0; k < n; k++) dst[k] = src[inv[k]];
        wht_normalized_chunks(dst, n);
        double *s2 = src; src = dst; dst = s2;
    }
    if (src != x) memcpy(x, src, sizeof(double) * n);
    free(scratch); free(inv);
}
/* TurboQuantizer::dequantize::<f64> (quantization.rs:321-376): out[padded_dim], still in the rotated space */
void qo_tq_dequantize(const qo_tq *t, const uint8_t *vec, double *out) {
    const uint32_t pd = t->padded_dim, bs = (uint32_t)bit_size(t->bits), code_bytes = pd * bs / 8;
    int nc;
    const float *centroids = centroids_of(t->bits, &nc);
    float sf;
    memcpy(&sf, vec + code_bytes, 4);
    const double scaling_factor = (double)sf;
    for (uint32_t i = 0; i < pd; i++) out[i] = (double)centroids[code_at(vec, i, bs)];
    double recovered_l2;
    if (t->distance == QO_DOT || t->distance == QO_COSINE) {
        double sq = 0.0;
        for (uint32_t i = 0; i < pd; i++) {
            const double r = t->shift ? out[i] / (double)t->scale[i] - (double)t->shift[i] : out[i];
            sq += r * r;
        }
        recovered_l2 = scaling_factor * sqrt(sq);
    } else if (t->distance == QO_EUCLID) {
        float l2;
        memcpy(&l2, vec + code_bytes + 4, 4);
        recovered_l2 = (double)l2;
    } else {
        recovered_l2 = scaling_factor;
    }
    const double scale = recovered_l2 / sqrt((double)pd);
    for (uint32_t i = 0; i < pd; i++) {
        const double r = t->shift ? out[i] / (double)t->scale[i] - (double)t->shift[i] : out[i];
        out[i] = r * scale;
    }
}

float qo_tq_score_precomputed(const qo_tq *t, const qo_tq_query *e, const uint8_t *vec) {
    const uint32_t code_bytes = t->padded_dim * (uint32_t)bit_size(t->bits) / 8;
    if (t->distance == QO_MANHATTAN) {     /* :596-607: dequantize, rotate back, sum of |q - v| as f32 over the query's coordinates, in order */
        double *deq = (double *)malloc(sizeof(double) * (t->padded_dim ? t->padded_dim : 1));
        qo_tq_dequantize(t, vec, deq);
        qo_tq_rotate_inverse(t, deq);
        float sum = 0.0f;
        const uint32_t n = t->dim < t->padded_dim ? t->dim : t->padded_dim;
        for (uint32_t i = 0; i < n; i++) sum += (float)fabs((double)e->raw[i] - deq[i]);
        free(deq);
        return sum;
    }
    float scaling_factor, l2 = 0.0f;
    memcpy(&scaling_factor, vec + code_bytes, 4);
    const float dot = tq_raw_dot(t, e, vec) + e->ec_correction;   /* 0.0 without TQ+ */
    if (t->distance == QO_EUCLID) {
        memcpy(&l2, vec + code_bytes + 4, 4);
        const float ql = e->l2_norm;
        return ql * ql + l2 * l2 - 2.0f * dot * scaling_factor;
    }
    return dot * scaling_factor;
}

float qo_tq_score_symmetric(const qo_tq *t, const uint8_t *v1, const uint8_t *v2) {
    const uint32_t pd = t->padded_dim, bs = (uint32_t)bit_size(t->bits), code_bytes = pd * bs / 8;
    if (t->distance == QO_MANHATTAN) {     /* :429-440: both dequantized, ONE inverse rotation of the difference, sum of |x| as f32 over padded_dim */
        double *d1 = (double *)malloc(sizeof(double) * (pd ? pd : 1)), *d2 = (double *)malloc(sizeof(double) * (pd ? pd : 1));
        qo_tq_dequantize(t, v1, d1);
        qo_tq_dequantize(t, v2, d2);
        for (uint32_t i = 0; i < pd; i++) d2[i] = d1[i] - d2[i];
        qo_tq_rotate_inverse(t, d2);
        float sum = 0.0f;
        for (uint32_t i = 0; i < pd; i++) sum += (float)fabs(d2[i]);
        free(d1); free(d2);
        return sum;
    }
    float raw_dot;
    if (t->shift && bs != 1) {      /* score_symmetric_ec (:447-494): weighted integer dot / (weight_scale * CODEBOOK_SCALE^2) + xm_a + xm_b - <M, M> */
        const uint8_t *book = t->bits == TQ_BITS4 ? CODEBOOK_U8_4BIT : CODEBOOK_U8_2BIT;
        int64_t acc = 0;
        for (uint32_t i = 0; i < pd; i++)
            acc += ((int64_t)book[code_at(v1, i, bs)] - TQ_CODEBOOK_OFFSET) * ((int64_t)book[code_at(v2, i, bs)] - TQ_CODEBOOK_OFFSET) *
                   (int64_t)t->d_prime_sq_i16[i];
        const float codebook_scale = 128.0f / (t->bits == TQ_BITS4 ? 2.733f : 1.510f);
        const float codebook_scale_sq = codebook_scale * codebook_scale;
        const float weighted = (float)acc / (t->weight_scale * codebook_scale_sq);
        const uint32_t xoff = code_bytes + (t->distance == QO_EUCLID ? 8 : 4);
        float xa, xb;
        memcpy(&xa, v1 + xoff, 4);
        memcpy(&xb, v2 + xoff, 4);
        raw_dot = weighted + xa + xb - t->mm_const;
    } else if (bs == 1) {
        uint64_t popcnt = 0;
        for (uint32_t i = 0; i < code_bytes; i++) popcnt += (uint64_t)__builtin_popcount((unsigned)(v1[i] ^ v2[i]));
        const int64_t sign_sum = (int64_t)code_bytes * 8 - 2 * (int64_t)popcnt;
        const float centroid_sq = 0.7978846f * 0.7978846f;
        raw_dot = centroid_sq * (float)sign_sum;
    } else {
        const uint8_t *book = t->bits == TQ_BITS4 ? CODEBOOK_U8_4BIT : CODEBOOK_U8_2BIT;
        int64_t acc = 0;
        for (uint32_t i = 0; i < pd; i++)
            acc += ((int64_t)book[code_at(v1, i, bs)] - TQ_CODEBOOK_OFFSET) * ((int64_t)book[code_at(v2, i, bs)] - TQ_CODEBOOK_OFFSET);
        const float codebook_scale = 128.0f / (t->bits == TQ_BITS4 ? 2.733f : 1.510f);
        raw_dot = (float)acc / (codebook_scale * codebook_scale);
    }
    float s1, s2;
    memcpy(&s1, v1 + code_bytes, 4);
    memcpy(&s2, v2 + code_bytes, 4);
    if (t->distance == QO_EUCLID) {
        float a, b;
        memcpy(&a, v1 + code_bytes + 4, 4);
        memcpy(&b, v2 + code_bytes + 4, 4);
        return a * a + b * b - 2.0f * s1 * s2 * raw_dot;
    }
    return raw_dot * s1 * s2;
}


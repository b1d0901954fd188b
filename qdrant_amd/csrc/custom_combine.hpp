// custom_combine.hpp — Query::score_by of the custom queries, shared by the similarity-matrix kernel (custom_query.hip) and the hop policy of
// the HNSW walk (hnsw.hpp HopCustom).  One formula source: `sim(e)` returns the similarity of the query's e-th example (flat_iter() order:
// reco: positives then negatives; discover / feedback: target, then (positive, negative) per pair; context: pairs) with the point.
//   RecoBestScoreQuery::score_by   vector_storage/query/reco_query.rs:68-92    (max by total_cmp, scaled_fast_sigmoid)
//   RecoSumScoresQuery::score_by   reco_query.rs:114-131                        (sequential f32 sums, pos - neg)
//   DiscoverQuery::score_by        discover_query.rs:45-73 (+ ContextPair::rank_by context_query.rs:38-45)
//   ContextQuery::score_by         context_query.rs:53-62, 112-118              (sum of fast_sigmoid(min(pos - neg - EPSILON, 0)))
//   FeedbackQuery::score_by        feedback_query.rs:198-226                    (a * sim(target) + sum pc_i * (sim(pos_i) - sim(neg_i)))
//   fast_sigmoid / scaled_fast_sigmoid  lib/common/common/src/math.rs:7-18
#pragma once
#include "common.hpp"

namespace qmx {

__device__ __forceinline__ int f32_total_cmp(float a, float b) {   // f32::total_cmp
    int32_t x = __float_as_int(a), y = __float_as_int(b);
    x ^= (int32_t)(((uint32_t)(x >> 31)) >> 1);
    y ^= (int32_t)(((uint32_t)(y >> 31)) >> 1);
    return x < y ? -1 : x > y ? 1 : 0;
}
__device__ __forceinline__ float fast_sigmoid(float x) { return x / (1.0f + __builtin_fabsf(x)); }
__device__ __forceinline__ float scaled_fast_sigmoid(float x) { return 0.5f * (fast_sigmoid(x) + 1.0f); }

// coefs: the feedback query's [a, partial_computation_0, ...] (already offset by coef_first), unused by the other kinds.
// ONE call site of sim (one loop over the examples, the kind decides what an example does to the state): the walk's policies inline a whole row scorer
// there.  The arithmetic is the reference's, operation for operation:
//   best score : max by total_cmp over the positives, then over the negatives                         (reco_query.rs:68-92)
//   sum scores : pos += / neg +=, from 0.0, in order                                                  (reco_query.rs:114-131)
//   discover   : rank += total_cmp(positive, negative) per pair; + scaled_fast_sigmoid(target)        (discover_query.rs:45-73)
//   feedback   : score = a * sim(target); score += pc_i * (sim(pos_i) - sim(neg_i)) pair by pair      (feedback_query.rs:198-226)
//   context    : sum += fast_sigmoid(min(pos - neg - EPSILON, 0)) per pair                            (context_query.rs:53-62, 112-118)
template <class Sim>
__device__ __forceinline__ float custom_score_by(uint32_t kind, uint32_t n_a, uint32_t n_b, const float *coefs, Sim sim) {
    const uint32_t ne = kind <= QMX_CUSTOM_RECO_SUM_SCORES ? n_a + n_b : n_a + 2 * n_b;
    const uint32_t lead = kind == QMX_CUSTOM_CONTEXT ? 0u : 1u;       // discover / feedback: example 0 is the target, pairs follow
    float a0 = kind == QMX_CUSTOM_RECO_BEST_SCORE ? -__builtin_inff() : 0.0f;       // max_pos | pos | target | score | sum
    float a1 = kind == QMX_CUSTOM_RECO_BEST_SCORE ? -__builtin_inff() : 0.0f;       // max_neg | neg
    float held = 0.0f;                                                             // the positive of the pair being read
    int32_t rank = 0;
    for (uint32_t e = 0; e < ne; ++e) {
        const float v = sim(e);
        if (kind == QMX_CUSTOM_RECO_BEST_SCORE) {
            if (e < n_a) { if (f32_total_cmp(v, a0) > 0) a0 = v; }
            else if (f32_total_cmp(v, a1) > 0) a1 = v;
        } else if (kind == QMX_CUSTOM_RECO_SUM_SCORES) {
            if (e < n_a) a0 += v;
            else a1 += v;
        } else if (e < lead) {
            a0 = kind == QMX_CUSTOM_FEEDBACK ? coefs[0] * v : v;
        } else if (((e - lead) & 1u) == 0) {
            held = v;
        } else if (kind == QMX_CUSTOM_DISCOVER) {
            rank += f32_total_cmp(held, v);
        } else if (kind == QMX_CUSTOM_FEEDBACK) {
            const float delta = held - v;
            a0 += coefs[1 + (e - lead) / 2] * delta;
        } else {
            const float difference = held - v - 1.1920929e-07f;   // ScoreType::EPSILON
            a0 += fast_sigmoid(__builtin_fminf(difference, 0.0f));
        }
    }
    switch (kind) {
        case QMX_CUSTOM_RECO_BEST_SCORE: return a0 > a1 ? scaled_fast_sigmoid(a0) : -scaled_fast_sigmoid(a1);
        case QMX_CUSTOM_RECO_SUM_SCORES: return a0 - a1;
        case QMX_CUSTOM_DISCOVER: return (float)rank + scaled_fast_sigmoid(a0);
        default: return a0;      // feedback, context
    }
}

}  // namespace qmx

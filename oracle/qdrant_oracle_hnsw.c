/*
 * qdrant_oracle_hnsw.c — CPU restatement of the HNSW graph (build + search), of the scorer façade
 * that drives it, and of the cross-segment result merge of qdrant v1.19.0.
 *
 * TEST INFRASTRUCTURE ONLY (see qdrant_oracle.h).  Follows
 *   lib/segment/src/index/hnsw_index/graph_layers.rs          search_on_level :108-149, search_entry :247-277,
 *                                                             search_entry_on_level :279-317, search :530-562
 *   lib/segment/src/index/hnsw_index/search_context.rs        :8-40
 *   lib/segment/src/index/hnsw_index/graph_layers_builder.rs  get_random_layer :388-396, link_new_point :417-474,
 *                                                             link_new_point_on_level :502-530, link_with_heuristic :532-556,
 *                                                             link_without_heuristic :558-577, for_each_link (ready_list) :56-68
 *   lib/segment/src/index/hnsw_index/links_container.rs       fill_from_sorted_with_heuristic :47-71, connect :74-103,
 *                                                             connect_with_heuristic_simple :106-132 (the reference's own
 *                                                             test :394-453 asserts it equals connect_with_heuristic)
 *   lib/segment/src/index/hnsw_index/entry_points.rs          new_point :46-94, get_entry_point :96-112
 *   lib/segment/src/index/hnsw_index/graph_links/serializer.rs :52-87,101-176 (plain layout)
 *   lib/segment/src/index/visited_pool.rs                     :9-80 (generation counters)
 *   lib/shard/src/search_result_aggregator.rs                 :11-47, 50-121
 * Rust std BinaryHeap (candidates max-heap) is restated from its published algorithm
 * (push = sift_up, pop = swap with last + sift_down_to_bottom + sift_up), like the bounded queue.
 *
 * Pinned by the reference's literal test `test_connect_new_point` (links_container.rs:312-391:
 * heuristic -> [1, 3, 6], connect -> [1..6]) in tests/test_oracle_hnsw.py.  Unpinned: random levels
 * (the reference uses rand's thread rng) and parallel insertion order — parity of the SEARCH is
 * defined on a given graph.
 */
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "qdrant_oracle.h"

static inline int get_bit(const uint64_t *bits, size_t i) { return (int)((bits[i >> 6] >> (i & 63)) & 1); }
static inline size_t elem_size(int dtype) { return dtype == QO_F32 ? 4 : dtype == QO_F16 ? 2 : 1; }

/* ------------------------------------------------------------------------------------------
 * scorer façade
 * ---------------------------------------------------------------------------------------- */
int qo_scorer_check_vector(const qo_scorer *s, uint32_t id) { /* NotDeletedChecker::check, raw_scorer.rs:596-603 */
    const qo_storage *st = s->st;
    int vec_del = (st->vec_deleted && id < st->n_vec_bits) ? get_bit(st->vec_deleted, id) : 0;
    int pt_del;
    if (st->point_deleted) pt_del = id < st->n_point_bits ? get_bit(st->point_deleted, id) : 1;
    else pt_del = id < st->n ? 0 : 1;
    return !vec_del && !pt_del;
}

float qo_scorer_score_point(const qo_scorer *s, uint32_t id) {
    float out = 0.f;
    if (s->kind == 6) {   /* query.score_by(|example| similarity(example, point)): the example similarities in flat_iter() order, then the query's own formula */
        const uint32_t ne = s->cq_kind <= 1 ? s->cq_n_a + s->cq_n_b : s->cq_n_a + 2 * s->cq_n_b;
        float sims[256];
        float *buf = ne <= 256 ? sims : (float *)malloc((size_t)ne * sizeof(float));
        for (uint32_t e = 0; e < ne; e++) buf[e] = qo_scorer_score_point(&s->cq_examples[e], id);
        const float r = s->cq_kind == 4 ? qo_custom_feedback(s->cq_n_b, buf, s->cq_coefs) : qo_custom_combine((int)s->cq_kind, s->cq_n_a, s->cq_n_b, buf);
        if (buf != sims) free(buf);
        return r;
    }
    if (s->kind == 4) {   /* sum over the query's inner vectors (from 0.0) of the max over the point's inner vectors (`if max_sim < sim`, from -inf) */
        float sum = 0.0f;
        for (uint32_t t = 0; t < s->mv_n_tokens; t++) {
            float max_sim = -INFINITY;
            for (uint64_t b = s->mv_offsets[id]; b < s->mv_offsets[id + 1]; b++) {
                const float sim = qo_scorer_score_point(&s->mv_tokens[t], (uint32_t)b);
                if (max_sim < sim) max_sim = sim;
            }
            sum += max_sim;
        }
        return sum;
    }
    if (s->kind == 5) {   /* EncodedVectorsTQ::score_point: score_precomputed, then `invert` */
        const float sc = qo_tq_score_precomputed(s->tq, s->tq_query, s->tq_rows + (size_t)id * qo_tq_quantized_size(s->tq));
        return s->tq_invert ? -sc : sc;
    }
    switch (s->kind) {
        case 0: qo_score_points(s->st, s->query, &id, 1, &out); return out;
        case 1: return qo_sq_score(s->sq, s->sq_query, s->sq_query_offset,
                                   s->sq_rows + (size_t)id * (4 + s->sq->actual_dim), s->isa);
        case 2: return qo_pq_score(s->pq, s->pq_lut, s->pq_codes + (size_t)id * s->pq->m, s->isa);
        default: return qo_bq_score(s->bq_distance, s->bq_invert, s->bq_dim, s->bq_query, s->bq_rows + (size_t)id * qo_bq_row_bytes(s->bq_dim));
    }
}

float qo_scorer_score_internal(const qo_scorer *s, uint32_t a, uint32_t b) {
    float out = 0.f;
    if (s->kind == 4) {   /* score_internal_max_similarity: every inner vector of a against the inner vectors of b (`if sim > max_sim`) */
        float sum = 0.0f;
        for (uint64_t i = s->mv_offsets[a]; i < s->mv_offsets[a + 1]; i++) {
            float max_sim = -INFINITY;
            for (uint64_t j = s->mv_offsets[b]; j < s->mv_offsets[b + 1]; j++) {
                const float sim = qo_scorer_score_internal(&s->mv_tokens[0], (uint32_t)i, (uint32_t)j);
                if (sim > max_sim) max_sim = sim;
            }
            sum += max_sim;
        }
        return sum;
    }
    if (s->kind == 5) {   /* EncodedVectorsTQ::score_internal: score_symmetric, then `invert` */
        const size_t rb = qo_tq_quantized_size(s->tq);
        const float sc = qo_tq_score_symmetric(s->tq, s->tq_rows + (size_t)a * rb, s->tq_rows + (size_t)b * rb);
        return s->tq_invert ? -sc : sc;
    }
    switch (s->kind) {
        case 0: { /* MetricQueryScorer::score_internal: similarity(get_dense(a), get_dense(b)), metric_query_scorer.rs:94-99 */
            const char *ra = (const char *)s->st->rows + (size_t)a * s->st->dim * elem_size(s->st->dtype);
            qo_score_points(s->st, ra, &b, 1, &out);
            return out;
        }
        case 1: {
            const size_t rb = 4 + s->sq->actual_dim;
            return qo_sq_score_internal(s->sq, s->sq_rows + (size_t)a * rb, s->sq_rows + (size_t)b * rb, s->isa);
        }
        case 2: return qo_pq_score_internal(s->pq, s->pq_codes + (size_t)a * s->pq->m, s->pq_codes + (size_t)b * s->pq->m);
        default: {  /* EncodedVectorsBin::score_internal :892-917 */
            const size_t rb = qo_bq_row_bytes(s->bq_dim);
            return qo_bq_score(s->bq_distance, s->bq_invert, s->bq_dim, s->bq_rows + (size_t)a * rb, s->bq_rows + (size_t)b * rb);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * BatchResultAggregator for disjoint or overlapping id spaces, all versions equal
 * ---------------------------------------------------------------------------------------- */
void qo_merge_topk(const qo_scored_point *lists, const uint32_t *counts, const uint32_t *idx_base,
                   uint32_t n_lists, uint32_t nq, uint32_t k, qo_scored_point *out, uint32_t *out_counts) {
    uint32_t *seen = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)n_lists * k + 1));
    for (uint32_t qi = 0; qi < nq; qi++) {
        qo_topk *q = qo_topk_new(k);          /* SearchResultAggregator::new(limit) */
        size_t n_seen = 0;
        for (uint32_t l = 0; l < n_lists; l++) {
            const uint32_t cnt = counts ? counts[(size_t)l * nq + qi] : k;
            const qo_scored_point *src = lists + ((size_t)l * nq + qi) * k;
            for (uint32_t i = 0; i < cnt && i < k; i++) {
                const uint32_t id = src[i].idx + (idx_base ? idx_base[l] : 0u);
                int dup = 0;                  /* `if self.seen.insert(point.id)` :33-36 */
                for (size_t j = 0; j < n_seen; j++) if (seen[j] == id) { dup = 1; break; }
                if (dup) continue;
                seen[n_seen++] = id;
                qo_topk_push(q, id, src[i].score);
            }
        }
        out_counts[qi] = (uint32_t)qo_topk_into_sorted(q, out + (size_t)qi * k);
        qo_topk_free(q);
    }
    free(seen);
}

/* ------------------------------------------------------------------------------------------
 * BinaryHeap<ScoredPointOffset> (max-heap by OrderedFloat(score)) — SearchContext::candidates
 * ---------------------------------------------------------------------------------------- */
typedef struct { qo_scored_point *d; size_t len, cap; } maxheap;
static inline int sp_cmp(const qo_scored_point *a, const qo_scored_point *b) { return qo_ordered_float_cmp(a->score, b->score); }
static void mh_sift_up(qo_scored_point *d, size_t start, size_t pos) {
    qo_scored_point elt = d[pos];
    while (pos > start) {
        size_t parent = (pos - 1) / 2;
        if (sp_cmp(&elt, &d[parent]) <= 0) break;
        d[pos] = d[parent];
        pos = parent;
    }
    d[pos] = elt;
}
static void mh_push(maxheap *h, qo_scored_point v) {
    if (h->len == h->cap) { h->cap = h->cap ? h->cap * 2 : 64; h->d = (qo_scored_point *)realloc(h->d, h->cap * sizeof(*h->d)); }
    h->d[h->len] = v;
    mh_sift_up(h->d, 0, h->len);
    h->len++;
}
static int mh_pop(maxheap *h, qo_scored_point *out) {
    if (h->len == 0) return 0;
    qo_scored_point item = h->d[--h->len];
    if (h->len > 0) {
        qo_scored_point t = h->d[0]; h->d[0] = item; item = t;
        /* sift_down_to_bottom(0) */
        const size_t end = h->len;
        size_t pos = 0, child = 1;
        qo_scored_point elt = h->d[0];
        while (end >= 2 && child <= end - 2) {
            child += (sp_cmp(&h->d[child], &h->d[child + 1]) <= 0) ? 1 : 0;
            h->d[pos] = h->d[child];
            pos = child;
            child = 2 * pos + 1;
        }
        if (child == end - 1) { h->d[pos] = h->d[child]; pos = child; }
        h->d[pos] = elt;
        mh_sift_up(h->d, 0, pos);
    }
    *out = item;
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * graph
 * ---------------------------------------------------------------------------------------- */
struct qo_hnsw {
    uint32_t n, m, m0, ef_construct, entry_points_num;
    int use_heuristic;
    uint32_t *level;        /* [n] highest level of each point (links_layers[p].len() - 1) */
    uint32_t **links;       /* [n] -> m0 + level * m ids */
    uint16_t **lens;        /* [n] -> level + 1 counts */
    _Atomic uint8_t *ready; /* ready_list */
    _Atomic uint32_t max_level;
    /* EntryPoints */
    uint32_t *ep_ids, *ep_levels; uint32_t ep_len, ep_cap;
    uint32_t *xp_ids, *xp_levels; uint32_t xp_len;      /* extra_entry_points: the `entry_points_num` highest seen */
    pthread_mutex_t ep_lock;
    pthread_spinlock_t *locks;                          /* per point, parallel build only */
    const qo_storage *st;                               /* build only */
};

static inline uint32_t level_m(const qo_hnsw *g, uint32_t level) { return level == 0 ? g->m0 : g->m; }   /* mod.rs:38-40 */
static inline uint32_t *links_ptr(const qo_hnsw *g, uint32_t p, uint32_t level) {
    return g->links[p] + (level == 0 ? 0 : g->m0 + (size_t)(level - 1) * g->m);
}

uint32_t qo_hnsw_point_level(const qo_hnsw *g, uint32_t id) { return g->level[id]; }
uint32_t qo_hnsw_max_level(const qo_hnsw *g) { return atomic_load(&g->max_level); }
uint32_t qo_hnsw_links(const qo_hnsw *g, uint32_t id, uint32_t level, uint32_t *out) {
    if (level > g->level[id]) return 0;
    const uint32_t len = g->lens[id][level];
    memcpy(out, links_ptr(g, id, level), sizeof(uint32_t) * len);
    return len;
}
uint32_t qo_hnsw_entry_points(const qo_hnsw *g, uint32_t *ids, uint32_t *levels, uint32_t cap) {
    for (uint32_t i = 0; i < g->ep_len && i < cap; i++) {
        if (ids) ids[i] = g->ep_ids[i];
        if (levels) levels[i] = g->ep_levels[i];
    }
    return g->ep_len;
}

/* EntryPoints::extra_entry_points in iter_unsorted order (entry_points.rs:13-15) */
uint32_t qo_hnsw_extra_entry_points(const qo_hnsw *g, uint32_t *ids, uint32_t *levels, uint32_t cap) {
    for (uint32_t i = 0; i < g->xp_len && i < cap; i++) {
        if (ids) ids[i] = g->xp_ids[i];
        if (levels) levels[i] = g->xp_levels[i];
    }
    return g->xp_len;
}

void qo_hnsw_free(qo_hnsw *g) {
    if (!g) return;
    for (uint32_t i = 0; i < g->n; i++) { free(g->links[i]); free(g->lens[i]); }
    free(g->links); free(g->lens); free(g->level); free((void *)g->ready);
    free(g->ep_ids); free(g->ep_levels); free(g->xp_ids); free(g->xp_levels);
    if (g->locks) { for (uint32_t i = 0; i < g->n; i++) pthread_spin_destroy(&g->locks[i]); free((void *)g->locks); }
    pthread_mutex_destroy(&g->ep_lock);
    free(g);
}

/* copies the links of (p, level) that are ready (graph_layers_builder.rs:56-68); locked in a parallel build */
static uint32_t read_links(const qo_hnsw *g, uint32_t p, uint32_t level, uint32_t *out, int only_ready) {
    if (g->locks) pthread_spin_lock(&g->locks[p]);
    const uint32_t len = g->lens[p][level];
    memcpy(out, links_ptr(g, p, level), sizeof(uint32_t) * len);
    if (g->locks) pthread_spin_unlock(&g->locks[p]);
    if (!only_ready) return len;
    uint32_t k = 0;
    for (uint32_t i = 0; i < len; i++) if (atomic_load(&g->ready[out[i]])) out[k++] = out[i];
    return k;
}

/* VisitedList: generation counters (visited_pool.rs:18-80) */
typedef struct { uint8_t cur; uint8_t *cnt; uint32_t n; } visited_t;
static void visited_init(visited_t *v, uint32_t n) { v->cur = 1; v->n = n; v->cnt = (uint8_t *)calloc(n ? n : 1, 1); }
static void visited_next(visited_t *v) {
    v->cur++;
    if (v->cur == 0) { v->cur = 1; memset(v->cnt, 0, v->n); }     /* next_iteration: wrap -> reset */
}
static inline int visited_check(const visited_t *v, uint32_t id) { return id < v->n && v->cnt[id] == v->cur; }
static inline int visited_check_update(visited_t *v, uint32_t id) { int was = v->cnt[id] == v->cur; v->cnt[id] = v->cur; return was; }

/* what a search needs from "the scorer of the query": either a qo_scorer (search) or the internal
 * scorer of point `self` during the build (score(id) = similarity(row self, row id)) */
typedef struct {
    const qo_scorer *s;       /* search */
    const qo_scorer *tmpl;    /* build: dense template (kind 0), query ignored */
    uint32_t self;
    uint64_t n_scored;
    /* qo_hnsw_search_traced: the candidates search_on_level pops and expands, in order (a count above the capacity = incomplete list) */
    qo_scored_point *trace;
    uint32_t trace_cap, trace_n;
} qscore;
static inline float qs_score(qscore *q, uint32_t id) {
    q->n_scored++;
    if (q->s) return qo_scorer_score_point(q->s, id);
    return qo_scorer_score_internal(q->tmpl, q->self, id);
}
static inline int qs_check(const qscore *q, uint32_t id) { return qo_scorer_check_vector(q->s ? q->s : q->tmpl, id); }

/* FilteredScorer::score_points (point_scorer.rs:265-280): keep ids passing the filters, truncate to limit */
static uint32_t filter_truncate(const qscore *q, uint32_t *ids, uint32_t n, uint32_t limit) {
    uint32_t k = 0;
    for (uint32_t i = 0; i < n; i++) if (qs_check(q, ids[i])) ids[k++] = ids[i];
    if (limit != 0 && k > limit) k = limit;
    return k;
}

/* SearchContext::process_candidate (search_context.rs:32-40) */
static void process_candidate(qo_topk *nearest, maxheap *cands, qo_scored_point sp) {
    qo_scored_point removed;
    const int r = qo_topk_push_ex(nearest, sp.idx, sp.score, &removed);
    const int was_added = (r == 0) || (removed.idx != sp.idx);
    if (was_added) mh_push(cands, sp);
}

/* GraphLayersBase::search_on_level (graph_layers.rs:108-149); returns the `nearest` queue */
static qo_topk *search_on_level(const qo_hnsw *g, qscore *q, qo_scored_point level_entry, uint32_t level, uint32_t ef,
                                visited_t *vis, int only_ready) {
    visited_next(vis);
    visited_check_update(vis, level_entry.idx);
    qo_topk *nearest = qo_topk_new(ef);
    maxheap cands = {0, 0, 0};
    process_candidate(nearest, &cands, level_entry);
    const uint32_t limit = level_m(g, level);
    uint32_t *ids = (uint32_t *)malloc(sizeof(uint32_t) * (g->m0 + g->m + 1));
    qo_scored_point cand;
    while (mh_pop(&cands, &cand)) {
        qo_scored_point worst;
        const float lower_bound = qo_topk_top(nearest, &worst) ? worst.score : -3.40282347e+38f;   /* ScoreType::min_value() */
        if (cand.score < lower_bound) break;
        if (q->trace) { if (q->trace_n < q->trace_cap) q->trace[q->trace_n] = cand; q->trace_n++; }
        uint32_t n = read_links(g, cand.idx, level, ids, only_ready), k = 0;
        for (uint32_t i = 0; i < n; i++) if (!visited_check(vis, ids[i])) ids[k++] = ids[i];
        k = filter_truncate(q, ids, k, limit);
        float scores[512];
        for (uint32_t i = 0; i < k; i++) scores[i] = qs_score(q, ids[i]);
        for (uint32_t i = 0; i < k; i++) {
            qo_scored_point sp = {ids[i], scores[i]};
            process_candidate(nearest, &cands, sp);
            visited_check_update(vis, ids[i]);
        }
    }
    free(ids);
    free(cands.d);
    return nearest;
}

/* GraphLayersBase::search_on_level_acorn (graph_layers.rs:154-243): the ACORN-1 walk.  Links whose point fails the filters are not
 * scored but EXPLORED: their own links (2 hops from the candidate) are offered too.  Two visited lists (hop1 / hop2), limits per
 * explored node (hop1_limit = hop2_limit = get_m(level)), scoring through score_points_unfiltered (point_scorer.rs:282-295). */
static qo_topk *search_on_level_acorn(const qo_hnsw *g, qscore *q, qo_scored_point level_entry, uint32_t level, uint32_t ef,
                                      visited_t *hop1_vis, visited_t *hop2_vis) {
    visited_next(hop1_vis);
    visited_next(hop2_vis);
    visited_check_update(hop1_vis, level_entry.idx);
    qo_topk *nearest = qo_topk_new(ef);
    maxheap cands = {0, 0, 0};
    process_candidate(nearest, &cands, level_entry);
    const uint32_t hop1_limit = level_m(g, level), hop2_limit = level_m(g, level);
    const uint32_t max_links = g->m0 + g->m + 1;
    uint32_t *links = (uint32_t *)malloc(sizeof(uint32_t) * max_links);
    uint32_t *links2 = (uint32_t *)malloc(sizeof(uint32_t) * max_links);
    uint32_t *to_explore = (uint32_t *)malloc(sizeof(uint32_t) * max_links);
    const size_t score_cap = (size_t)max_links * (max_links + 1);
    uint32_t *to_score = (uint32_t *)malloc(sizeof(uint32_t) * score_cap);
    float *scores = (float *)malloc(sizeof(float) * score_cap);
    qo_scored_point cand;
    while (mh_pop(&cands, &cand)) {
        qo_scored_point worst;
        const float lower_bound = qo_topk_top(nearest, &worst) ? worst.score : -3.40282347e+38f;
        if (cand.score < lower_bound) break;
        uint32_t n_explore = 0, n_score = 0;
        /* 1-hop neighbours :196-211 */
        const uint32_t n = read_links(g, cand.idx, level, links, 0);
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t hop1 = links[i];
            if (visited_check_update(hop1_vis, hop1)) continue;
            if (qs_check(q, hop1)) {
                to_score[n_score++] = hop1;
                if (n_score >= hop1_limit) break;
            } else {
                to_explore[n_explore++] = hop1;
            }
        }
        /* 2-hop neighbours :213-235 */
        for (uint32_t e = 0; e < n_explore; e++) {
            const uint32_t total_limit = n_score + hop2_limit;
            const uint32_t n2 = read_links(g, to_explore[e], level, links2, 0);
            for (uint32_t i = 0; i < n2; i++) {
                const uint32_t hop2 = links2[i];
                if (visited_check(hop1_vis, hop2) || visited_check_update(hop2_vis, hop2)) continue;
                if (qs_check(q, hop2)) {
                    visited_check_update(hop1_vis, hop2);
                    to_score[n_score++] = hop2;
                    if (n_score >= total_limit) break;
                }
            }
        }
        /* score_points_unfiltered + process_candidate in order :237-239 */
        for (uint32_t i = 0; i < n_score; i++) scores[i] = qs_score(q, to_score[i]);
        for (uint32_t i = 0; i < n_score; i++) {
            qo_scored_point sp = {to_score[i], scores[i]};
            process_candidate(nearest, &cands, sp);
        }
    }
    free(links); free(links2); free(to_explore); free(to_score); free(scores);
    free(cands.d);
    return nearest;
}

/* search_entry_on_level (graph_layers.rs:279-317) */
static qo_scored_point search_entry_on_level(const qo_hnsw *g, qscore *q, uint32_t entry, uint32_t level, int only_ready) {
    const uint32_t limit = level_m(g, level);
    uint32_t *links = (uint32_t *)malloc(sizeof(uint32_t) * (g->m0 + g->m + 1));
    qo_scored_point cur = {entry, qs_score(q, entry)};
    int changed = 1;
    while (changed) {
        changed = 0;
        uint32_t n = read_links(g, cur.idx, level, links, only_ready);
        n = filter_truncate(q, links, n, limit);
        float scores[512];
        for (uint32_t i = 0; i < n; i++) scores[i] = qs_score(q, links[i]);
        for (uint32_t i = 0; i < n; i++)
            if (scores[i] > cur.score) { changed = 1; cur.idx = links[i]; cur.score = scores[i]; }
    }
    free(links);
    return cur;
}

/* search_entry (graph_layers.rs:247-277): levels top_level .. target_level + 1 */
static qo_scored_point search_entry(const qo_hnsw *g, qscore *q, uint32_t entry, uint32_t top_level, uint32_t target_level,
                                    int only_ready) {
    qo_scored_point result = {entry, 0.f};
    int have = 0;
    uint32_t level_entry = entry;
    for (uint32_t level = top_level; level > target_level; level--) {
        result = search_entry_on_level(g, q, level_entry, level, only_ready);
        level_entry = result.idx;
        have = 1;
    }
    if (!have) { result.idx = entry; result.score = qs_score(q, entry); }
    return result;
}

/* EntryPoints::get_entry_point (entry_points.rs:96-112) */
static int get_entry_point(const qo_hnsw *g, const qscore *q, uint32_t *id, uint32_t *level) {
    for (uint32_t i = 0; i < g->ep_len; i++)
        if (qs_check(q, g->ep_ids[i])) { *id = g->ep_ids[i]; *level = g->ep_levels[i]; return 1; }
    int found = 0;
    for (uint32_t i = 0; i < g->xp_len; i++)      /* max_by_key(level): the last maximal element wins */
        if (qs_check(q, g->xp_ids[i]) && (!found || g->xp_levels[i] >= *level)) { *id = g->xp_ids[i]; *level = g->xp_levels[i]; found = 1; }
    return found;
}
static void extra_push(qo_hnsw *g, uint32_t id, uint32_t level) {   /* FixedLengthPriorityQueue<EntryPoint>, ordered by level */
    if (g->entry_points_num == 0) return;
    if (g->xp_len < g->entry_points_num) { g->xp_ids[g->xp_len] = id; g->xp_levels[g->xp_len] = level; g->xp_len++; return; }
    uint32_t lo = 0;
    for (uint32_t i = 1; i < g->xp_len; i++) if (g->xp_levels[i] < g->xp_levels[lo]) lo = i;
    if (g->xp_levels[lo] < level) { g->xp_ids[lo] = id; g->xp_levels[lo] = level; }
}
/* EntryPoints::new_point (entry_points.rs:46-94) */
static void entry_new_point(qo_hnsw *g, const qscore *q, uint32_t new_point, uint32_t level) {
    for (uint32_t i = 0; i < g->ep_len; i++) {
        if (!qs_check(q, g->ep_ids[i])) continue;
        if (g->ep_levels[i] >= level) {
            extra_push(g, new_point, level);
        } else {
            const uint32_t oid = g->ep_ids[i], olv = g->ep_levels[i];
            g->ep_ids[i] = new_point; g->ep_levels[i] = level;
            extra_push(g, oid, olv);
        }
        return;
    }
    if (g->ep_len == g->ep_cap) {
        g->ep_cap = g->ep_cap ? g->ep_cap * 2 : 8;
        g->ep_ids = (uint32_t *)realloc(g->ep_ids, sizeof(uint32_t) * g->ep_cap);
        g->ep_levels = (uint32_t *)realloc(g->ep_levels, sizeof(uint32_t) * g->ep_cap);
    }
    g->ep_ids[g->ep_len] = new_point; g->ep_levels[g->ep_len] = level; g->ep_len++;
}

/* ---- LinksContainer ------------------------------------------------------------------------ */
typedef float (*pair_score_fn)(void *ctx, uint32_t a, uint32_t b);

/* fill_from_sorted_with_heuristic (links_container.rs:47-71); returns the new length */
static uint32_t heuristic_fill(uint32_t *links, const qo_scored_point *cand, uint32_t n_cand, uint32_t lm,
                               pair_score_fn score, void *ctx) {
    uint32_t len = 0;
    if (lm == 0) return 0;
    for (uint32_t c = 0; c < n_cand; c++) {
        int skip = 0;
        for (uint32_t e = 0; e < len; e++)
            if (score(ctx, cand[c].idx, links[e]) > cand[c].score) { skip = 1; break; }
        if (skip) continue;
        links[len++] = cand[c].idx;
        if (len >= lm) break;
    }
    return len;
}
/* f32::total_cmp */
static inline int total_cmp(float a, float b) {
    int32_t x, y;
    memcpy(&x, &a, 4); memcpy(&y, &b, 4);
    x ^= (int32_t)(((uint32_t)(x >> 31)) >> 1);
    y ^= (int32_t)(((uint32_t)(y >> 31)) >> 1);
    return x < y ? -1 : x > y ? 1 : 0;
}
/* connect_with_heuristic_simple (links_container.rs:106-132); returns the new length */
static uint32_t connect_with_heuristic(uint32_t *links, uint32_t len, uint32_t new_point, uint32_t target, uint32_t lm,
                                       pair_score_fn score, void *ctx) {
    if (lm == 0) return len;
    if (len < lm) { links[len] = new_point; return len + 1; }
    qo_scored_point cand[513];
    for (uint32_t i = 0; i < len; i++) { cand[i].idx = links[i]; cand[i].score = score(ctx, target, links[i]); }
    cand[len].idx = new_point; cand[len].score = score(ctx, target, new_point);
    /* sort_unstable_by(|a, b| b.score.total_cmp(&a.score)): descending; equal keys keep input order here */
    for (uint32_t i = 1; i <= len; i++) {
        qo_scored_point x = cand[i];
        uint32_t j = i;
        while (j > 0 && total_cmp(cand[j - 1].score, x.score) < 0) { cand[j] = cand[j - 1]; j--; }
        cand[j] = x;
    }
    return heuristic_fill(links, cand, len + 1, lm, score, ctx);
}
/* connect (links_container.rs:74-103); returns the new length */
static uint32_t connect_plain(uint32_t *links, uint32_t len, uint32_t new_point, uint32_t target, uint32_t lm,
                              pair_score_fn score, void *ctx) {
    const float new_to_target = score(ctx, target, new_point);
    uint32_t at = len;
    for (uint32_t i = 0; i < len; i++) {
        const float target_to_link = score(ctx, target, links[i]);
        if (target_to_link < new_to_target) { at = i; break; }
    }
    if (len < lm) {
        memmove(links + at + 1, links + at, sizeof(uint32_t) * (len - at));
        links[at] = new_point;
        return len + 1;
    } else if (at != len) {
        len--;                                                  /* links.pop() */
        memmove(links + at + 1, links + at, sizeof(uint32_t) * (len - at));
        links[at] = new_point;
        return len + 1;
    }
    return len;
}

typedef struct { const float *table; uint32_t n; } table_ctx;
static float table_score(void *c, uint32_t a, uint32_t b) { const table_ctx *t = (const table_ctx *)c; return t->table[(size_t)a * t->n + b]; }
uint32_t qo_links_heuristic(const qo_scored_point *sorted_candidates, uint32_t n_cand, uint32_t lm, const float *score_table,
                            uint32_t n, uint32_t *out_links) {
    table_ctx t = {score_table, n};
    return heuristic_fill(out_links, sorted_candidates, n_cand, lm, table_score, &t);
}
uint32_t qo_links_connect(uint32_t *links, uint32_t len, uint32_t new_point, uint32_t target, uint32_t lm,
                          const float *score_table, uint32_t n) {
    table_ctx t = {score_table, n};
    return connect_plain(links, len, new_point, target, lm, table_score, &t);
}

uint32_t qo_links_connect_heuristic(uint32_t *links, uint32_t len, uint32_t new_point, uint32_t target, uint32_t lm,
                                    const float *score_table, uint32_t n) {
    table_ctx t = {score_table, n};
    return connect_with_heuristic(links, len, new_point, target, lm, table_score, &t);
}

static float internal_score(void *c, uint32_t a, uint32_t b) { return qo_scorer_score_internal((const qo_scorer *)c, a, b); }

/* ---- GraphLayersBuilder::link_new_point (graph_layers_builder.rs:417-474) --------------------- */
static void link_new_point(qo_hnsw *g, const qo_scorer *tmpl, uint32_t p, visited_t *vis) {
    qscore q = {NULL, tmpl, p, 0};
    /* FilteredScorer::new_internal (point_scorer.rs:183-218): a quantized storage that cannot rebuild a query from a stored row
     * (EncodedVectorsPQ::encode_internal_vector -> None) scores the searches of this insertion through
     * quantized_vectors.raw_scorer(ORIGINAL vector of p) = the LUT of the preprocessed original (quantized_query_scorer.rs:39-41);
     * score_internal (entry score below, the heuristic, the back links) stays the storage's symmetric one */
    qo_scorer own;
    float *lut = NULL;
    if (tmpl->kind == 2) {
        const qo_storage *st = tmpl->st;
        float *qv = (float *)malloc(sizeof(float) * st->dim);
        qo_preprocess_f32(st->distance, (const float *)st->rows + (size_t)p * st->dim, qv, st->dim);
        lut = (float *)malloc(sizeof(float) * (size_t)tmpl->pq->m * tmpl->pq->n_centroids);
        qo_pq_encode_query(tmpl->pq, qv, lut);
        free(qv);
        own = *tmpl;
        own.pq_lut = lut;
        q.s = &own;
    }
    qo_tq_query *tq_query = NULL;
    if (tmpl->kind == 5) {   /* EncodedVectorsTQ::encode_internal_vector -> None as well: the query of the ORIGINAL vector, score_symmetric elsewhere */
        const qo_storage *st = tmpl->st;
        float *qv = (float *)malloc(sizeof(float) * st->dim);
        qo_preprocess_f32(st->distance, (const float *)st->rows + (size_t)p * st->dim, qv, st->dim);
        tq_query = qo_tq_precompute_query(tmpl->tq, qv);
        free(qv);
        own = *tmpl;
        own.tq_query = tq_query;
        q.s = &own;
    }
    /* Multi-vector points over such inner rows (QuantizedMultivectorStorage::encode_internal_vector, quantized_multivector_storage/mod.rs:458-470: every
     * inner row's encode_internal_vector, `?` -> None as soon as one is None): the same fallback one level up - the multi-vector query scorer of the point's
     * ORIGINAL inner vectors (one inner query scorer per inner vector: a LUT / a precomputed TurboQuant query), score_internal_max_similarity elsewhere.
     * tmpl->mv_tokens[0] is the inner storage as a template; its `st` holds the original inner rows. */
    qo_scorer *mv_own = NULL;
    float **mv_luts = NULL;
    qo_tq_query **mv_tqq = NULL;
    uint32_t mv_n = 0;
    if (tmpl->kind == 4 && (tmpl->mv_tokens[0].kind == 2 || tmpl->mv_tokens[0].kind == 5)) {
        const qo_scorer *inner = &tmpl->mv_tokens[0];
        const qo_storage *st = inner->st;
        const uint64_t r0 = tmpl->mv_offsets[p], r1 = tmpl->mv_offsets[p + 1];
        mv_n = (uint32_t)(r1 - r0);
        mv_own = (qo_scorer *)malloc(sizeof(qo_scorer) * (mv_n ? mv_n : 1));
        mv_luts = (float **)calloc(mv_n ? mv_n : 1, sizeof(float *));
        mv_tqq = (qo_tq_query **)calloc(mv_n ? mv_n : 1, sizeof(qo_tq_query *));
        float *qv = (float *)malloc(sizeof(float) * st->dim);
        for (uint32_t t = 0; t < mv_n; t++) {
            qo_preprocess_f32(st->distance, (const float *)st->rows + (size_t)(r0 + t) * st->dim, qv, st->dim);
            mv_own[t] = *inner;
            if (inner->kind == 2) {
                mv_luts[t] = (float *)malloc(sizeof(float) * (size_t)inner->pq->m * inner->pq->n_centroids);
                qo_pq_encode_query(inner->pq, qv, mv_luts[t]);
                mv_own[t].pq_lut = mv_luts[t];
            } else {
                mv_tqq[t] = qo_tq_precompute_query(inner->tq, qv);
                mv_own[t].tq_query = mv_tqq[t];
            }
        }
        free(qv);
        own = *tmpl;
        own.mv_tokens = mv_own;
        own.mv_n_tokens = mv_n;
        q.s = &own;
    }
    const uint32_t level = g->level[p];
    uint32_t ep_id = 0, ep_level = 0;
    pthread_mutex_lock(&g->ep_lock);
    const int have_ep = get_entry_point(g, &q, &ep_id, &ep_level);
    pthread_mutex_unlock(&g->ep_lock);
    if (have_ep) {
        qo_scored_point level_entry;
        if (ep_level > level) level_entry = search_entry(g, &q, ep_id, ep_level, level, 1);
        else { level_entry.idx = ep_id; level_entry.score = qo_scorer_score_internal(tmpl, p, ep_id); }
        const uint32_t linking_level = level < ep_level ? level : ep_level;
        qo_scored_point *sorted = (qo_scored_point *)malloc(sizeof(qo_scored_point) * (g->ef_construct + 1));
        for (int32_t cl = (int32_t)linking_level; cl >= 0; cl--) {
            const uint32_t curr = (uint32_t)cl, lm = level_m(g, curr);
            /* link_new_point_on_level (:502-530) */
            qo_topk *nearest = search_on_level(g, &q, level_entry, curr, g->ef_construct, vis, 1);
            const size_t nn = qo_topk_len(nearest);
            const qo_scored_point *un = qo_topk_data(nearest);
            if (nn) {   /* iter_unsorted().max(): the last maximal element */
                size_t best = 0;
                for (size_t i = 1; i < nn; i++) if (qo_ordered_float_cmp(un[i].score, un[best].score) >= 0) best = i;
                level_entry = un[best];
            }
            if (g->use_heuristic) {   /* link_with_heuristic (:532-556) */
                const uint32_t ns = (uint32_t)qo_topk_into_sorted(nearest, sorted);
                uint32_t selected[512], n_sel;
                if (g->locks) pthread_spin_lock(&g->locks[p]);
                n_sel = heuristic_fill(links_ptr(g, p, curr), sorted, ns, lm, internal_score, (void *)tmpl);
                g->lens[p][curr] = (uint16_t)n_sel;
                memcpy(selected, links_ptr(g, p, curr), sizeof(uint32_t) * n_sel);
                if (g->locks) pthread_spin_unlock(&g->locks[p]);
                for (uint32_t i = 0; i < n_sel; i++) {
                    const uint32_t other = selected[i];
                    if (g->locks) pthread_spin_lock(&g->locks[other]);
                    g->lens[other][curr] = (uint16_t)connect_with_heuristic(links_ptr(g, other, curr), g->lens[other][curr], p, other,
                                                                            lm, internal_score, (void *)tmpl);
                    if (g->locks) pthread_spin_unlock(&g->locks[other]);
                }
            } else {                  /* link_without_heuristic (:558-577) */
                for (size_t i = 0; i < nn; i++) {
                    const uint32_t o = un[i].idx;
                    if (g->locks) pthread_spin_lock(&g->locks[p]);
                    g->lens[p][curr] = (uint16_t)connect_plain(links_ptr(g, p, curr), g->lens[p][curr], o, p, lm, internal_score, (void *)tmpl);
                    if (g->locks) pthread_spin_unlock(&g->locks[p]);
                    if (g->locks) pthread_spin_lock(&g->locks[o]);
                    g->lens[o][curr] = (uint16_t)connect_plain(links_ptr(g, o, curr), g->lens[o][curr], p, o, lm, internal_score, (void *)tmpl);
                    if (g->locks) pthread_spin_unlock(&g->locks[o]);
                }
            }
            qo_topk_free(nearest);
        }
        free(sorted);
    }
    atomic_store(&g->ready[p], 1);
    pthread_mutex_lock(&g->ep_lock);
    entry_new_point(g, &q, p, level);
    pthread_mutex_unlock(&g->ep_lock);
    free(lut);
    if (tq_query) qo_tq_query_free(tq_query);
    if (mv_own) {
        for (uint32_t t = 0; t < mv_n; t++) {
            free(mv_luts[t]);
            if (mv_tqq[t]) qo_tq_query_free(mv_tqq[t]);
        }
        free(mv_luts); free(mv_tqq); free(mv_own);
    }
}

static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static qo_hnsw *hnsw_alloc(const qo_storage *st, uint32_t m, uint32_t m0, uint32_t ef_construct, uint32_t entry_points_num,
                           int use_heuristic, uint64_t seed, int parallel) {
    qo_hnsw *g = (qo_hnsw *)calloc(1, sizeof(*g));
    g->n = (uint32_t)st->n; g->m = m; g->m0 = m0; g->ef_construct = ef_construct; g->entry_points_num = entry_points_num;
    g->use_heuristic = use_heuristic; g->st = st;
    g->level = (uint32_t *)calloc(g->n ? g->n : 1, sizeof(uint32_t));
    g->links = (uint32_t **)calloc(g->n ? g->n : 1, sizeof(uint32_t *));
    g->lens = (uint16_t **)calloc(g->n ? g->n : 1, sizeof(uint16_t *));
    g->ready = (_Atomic uint8_t *)calloc(g->n ? g->n : 1, 1);
    g->xp_ids = (uint32_t *)calloc(entry_points_num ? entry_points_num : 1, sizeof(uint32_t));
    g->xp_levels = (uint32_t *)calloc(entry_points_num ? entry_points_num : 1, sizeof(uint32_t));
    pthread_mutex_init(&g->ep_lock, NULL);
    /* level_factor = 1 / ln(max(m, 2)) ; level = round(-ln(U) * level_factor)  (:320, :388-396) */
    const double level_factor = 1.0 / log((double)(m > 2 ? m : 2));
    uint32_t maxl = 0;
    for (uint32_t i = 0; i < g->n; i++) {
        const uint64_t r = splitmix64(seed ^ (0xA0761D6478BD642Full * (i + 1)));
        const double u = ((double)(r >> 11) + 0.5) * (1.0 / 9007199254740992.0);   /* (0, 1) */
        double lv = round(-log(u) * level_factor);
        if (lv > 32) lv = 32;
        g->level[i] = (uint32_t)lv;
        if (g->level[i] > maxl) maxl = g->level[i];
        g->links[i] = (uint32_t *)malloc(sizeof(uint32_t) * (m0 + (size_t)g->level[i] * m + 1));
        g->lens[i] = (uint16_t *)calloc(g->level[i] + 1, sizeof(uint16_t));
    }
    atomic_store(&g->max_level, maxl);
    if (parallel) {
        g->locks = (pthread_spinlock_t *)malloc(sizeof(pthread_spinlock_t) * (g->n ? g->n : 1));
        for (uint32_t i = 0; i < g->n; i++) pthread_spin_init(&g->locks[i], PTHREAD_PROCESS_PRIVATE);
    }
    return g;
}

qo_hnsw *qo_hnsw_build(const qo_storage *st, uint32_t m, uint32_t m0, uint32_t ef_construct, uint32_t entry_points_num,
                       int use_heuristic, uint64_t seed) {
    qo_hnsw *g = hnsw_alloc(st, m, m0, ef_construct, entry_points_num, use_heuristic, seed, 0);
    qo_scorer tmpl;
    memset(&tmpl, 0, sizeof(tmpl));
    tmpl.kind = 0; tmpl.st = st;
    visited_t vis;
    visited_init(&vis, g->n);
    for (uint32_t p = 0; p < g->n; p++) {
        if (!qo_scorer_check_vector(&tmpl, p)) continue;     /* deleted points are never indexed (hnsw/build.rs:293-300) */
        link_new_point(g, &tmpl, p, &vis);
    }
    free(vis.cnt);
    return g;
}

/* hnsw/build.rs:334-341: a segment with quantized vectors builds its graph THROUGH the quantized scorer.  `tmpl` = the quantized
 * storage as a scorer template (kind 1 SQ / 2 PQ / 3 BQ; its `st` = the original dense storage: deleted flags, n, and — for PQ —
 * the original vectors the per-point LUTs are made of).  Sequential, like the reference's deterministic test builds. */
qo_hnsw *qo_hnsw_build_with(const qo_scorer *tmpl, uint32_t m, uint32_t m0, uint32_t ef_construct, uint32_t entry_points_num,
                            int use_heuristic, uint64_t seed) {
    qo_hnsw *g = hnsw_alloc(tmpl->st, m, m0, ef_construct, entry_points_num, use_heuristic, seed, 0);
    visited_t vis;
    visited_init(&vis, g->n);
    for (uint32_t p = 0; p < g->n; p++) {
        if (!qo_scorer_check_vector(tmpl, p)) continue;
        link_new_point(g, tmpl, p, &vis);
    }
    free(vis.cnt);
    return g;
}

typedef struct { qo_hnsw *g; const qo_scorer *tmpl; _Atomic uint32_t *next; uint32_t end; } build_arg;
static void *build_worker(void *a_) {
    build_arg *a = (build_arg *)a_;
    visited_t vis;
    visited_init(&vis, a->g->n);
    for (;;) {
        const uint32_t p = atomic_fetch_add(a->next, 1);
        if (p >= a->end) break;
        if (!qo_scorer_check_vector(a->tmpl, p)) continue;
        link_new_point(a->g, a->tmpl, p, &vis);
    }
    free(vis.cnt);
    return NULL;
}
qo_hnsw *qo_hnsw_build_parallel(const qo_storage *st, uint32_t m, uint32_t m0, uint32_t ef_construct, uint32_t entry_points_num,
                                int use_heuristic, uint64_t seed, int threads) {
    if (threads <= 1) return qo_hnsw_build(st, m, m0, ef_construct, entry_points_num, use_heuristic, seed);
    qo_hnsw *g = hnsw_alloc(st, m, m0, ef_construct, entry_points_num, use_heuristic, seed, 1);
    qo_scorer tmpl;
    memset(&tmpl, 0, sizeof(tmpl));
    tmpl.kind = 0; tmpl.st = st;
    const uint32_t single = g->n < 256 ? g->n : 256;          /* SINGLE_THREADED_HNSW_BUILD_THRESHOLD, hnsw.rs:38 */
    visited_t vis;
    visited_init(&vis, g->n);
    for (uint32_t p = 0; p < single; p++) if (qo_scorer_check_vector(&tmpl, p)) link_new_point(g, &tmpl, p, &vis);
    free(vis.cnt);
    _Atomic uint32_t next = single;
    build_arg arg = {g, &tmpl, &next, g->n};
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * threads);
    for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, build_worker, &arg);
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
    return g;
}

/* ---- plain GraphLinks layout (graph_links/serializer.rs:52-87,101-176) --------------------------- */
void qo_hnsw_export_plain(const qo_hnsw *g, uint32_t *n_levels, uint64_t *n_offsets, uint64_t *n_neighbors,
                          uint32_t *reindex, uint64_t *level_offsets, uint64_t *offsets, uint32_t *neighbors) {
    const uint32_t L = qo_hnsw_max_level(g) + 1;
    uint64_t *count_ge = (uint64_t *)calloc(L + 1, sizeof(uint64_t));   /* points with level >= l */
    for (uint32_t i = 0; i < g->n; i++) for (uint32_t l = 0; l <= g->level[i]; l++) count_ge[l]++;
    uint64_t total = 0, nn = 0;
    for (uint32_t l = 0; l < L; l++) total += count_ge[l];
    for (uint32_t i = 0; i < g->n; i++) for (uint32_t l = 0; l <= g->level[i]; l++) nn += g->lens[i][l];
    if (n_levels) *n_levels = L;
    if (n_offsets) *n_offsets = total + 1;
    if (n_neighbors) *n_neighbors = nn;
    if (!reindex || !level_offsets || !offsets || !neighbors) { free(count_ge); return; }
    /* back_index: points by descending level (sort_unstable_by_key(Reverse(levels)); ties by id here) */
    uint32_t *back = (uint32_t *)malloc(sizeof(uint32_t) * (g->n ? g->n : 1));
    {
        uint64_t *start = (uint64_t *)calloc(L + 1, sizeof(uint64_t));
        uint64_t acc = 0;
        for (int32_t l = (int32_t)L - 1; l >= 0; l--) { start[l] = acc; acc += count_ge[l] - (l + 1 < (int32_t)L ? count_ge[l + 1] : 0); }
        for (uint32_t i = 0; i < g->n; i++) back[start[g->level[i]]++] = i;
        free(start);
    }
    for (uint32_t i = 0; i < g->n; i++) reindex[back[i]] = i;
    uint64_t off = 0, slot = 0;
    for (uint32_t l = 0; l < L; l++) {
        level_offsets[l] = slot;
        for (uint64_t j = 0; j < count_ge[l]; j++) {
            const uint32_t id = l == 0 ? (uint32_t)j : back[j];
            offsets[slot++] = off;
            const uint32_t len = g->lens[id][l];
            memcpy(neighbors + off, links_ptr(g, id, l), sizeof(uint32_t) * len);
            off += len;
        }
    }
    level_offsets[L] = slot;
    offsets[slot] = off;
    free(back);
    free(count_ge);
}

/* ---- GraphLayers::load from the plain GraphLinks arrays (graph_links/view.rs) -------------------------------------
 * Lets the oracle WALK a graph it did not build (e.g. one built on the device) for parity checks of the search. */
qo_hnsw *qo_hnsw_import_plain(uint32_t n, uint32_t m, uint32_t m0, uint32_t n_levels, const uint32_t *reindex,
                              const uint64_t *level_offsets, const uint64_t *offsets, const uint32_t *neighbors,
                              const uint32_t *ep_ids, const uint32_t *ep_levels, uint32_t n_ep,
                              const uint32_t *xp_ids, const uint32_t *xp_levels, uint32_t n_xp) {
    qo_hnsw *g = (qo_hnsw *)calloc(1, sizeof(*g));
    g->n = n; g->m = m; g->m0 = m0; g->entry_points_num = n_xp; g->use_heuristic = 1;
    g->level = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
    g->links = (uint32_t **)calloc(n ? n : 1, sizeof(uint32_t *));
    g->lens = (uint16_t **)calloc(n ? n : 1, sizeof(uint16_t *));
    g->ready = (_Atomic uint8_t *)calloc(n ? n : 1, 1);
    pthread_mutex_init(&g->ep_lock, NULL);
    uint32_t maxl = 0;
    for (uint32_t i = 0; i < n; i++) {       /* level(p) = highest l whose slot range still holds reindex[p] */
        uint32_t lv = 0;
        for (uint32_t l = 1; l < n_levels; l++) if ((uint64_t)reindex[i] < level_offsets[l + 1] - level_offsets[l]) lv = l;
        g->level[i] = lv;
        if (lv > maxl) maxl = lv;
        g->links[i] = (uint32_t *)malloc(sizeof(uint32_t) * (m0 + (size_t)lv * m + 1));
        g->lens[i] = (uint16_t *)calloc(lv + 1, sizeof(uint16_t));
        for (uint32_t l = 0; l <= lv; l++) {
            const uint64_t slot = l == 0 ? i : level_offsets[l] + reindex[i];
            const uint64_t len = offsets[slot + 1] - offsets[slot];
            const uint32_t cap = level_m(g, l);
            g->lens[i][l] = (uint16_t)(len < cap ? len : cap);
            memcpy(links_ptr(g, i, l), neighbors + offsets[slot], sizeof(uint32_t) * g->lens[i][l]);
        }
        atomic_store(&g->ready[i], 1);
    }
    atomic_store(&g->max_level, maxl);
    g->ep_cap = n_ep ? n_ep : 1;
    g->ep_ids = (uint32_t *)malloc(sizeof(uint32_t) * g->ep_cap);
    g->ep_levels = (uint32_t *)malloc(sizeof(uint32_t) * g->ep_cap);
    memcpy(g->ep_ids, ep_ids, sizeof(uint32_t) * n_ep);
    memcpy(g->ep_levels, ep_levels, sizeof(uint32_t) * n_ep);
    g->ep_len = n_ep;
    g->xp_ids = (uint32_t *)calloc(n_xp ? n_xp : 1, sizeof(uint32_t));
    g->xp_levels = (uint32_t *)calloc(n_xp ? n_xp : 1, sizeof(uint32_t));
    memcpy(g->xp_ids, xp_ids, sizeof(uint32_t) * n_xp);
    memcpy(g->xp_levels, xp_levels, sizeof(uint32_t) * n_xp);
    g->xp_len = n_xp;
    return g;
}

/* ---- GraphLayers::search (graph_layers.rs:530-562) ------------------------------------------------ */
uint32_t qo_hnsw_search(const qo_hnsw *g, const qo_scorer *scorer, uint32_t top, uint32_t ef, qo_scored_point *out,
                        uint64_t *n_scored) {
    return qo_hnsw_search_algo(g, scorer, top, ef, 0, out, n_scored);
}
/* algorithm: 0 = SearchAlgorithm::Hnsw, 1 = SearchAlgorithm::Acorn (graph_layers.rs:550-559) */
/* GraphLayersWithVectors::search_on_level_with_vectors (graph_layers.rs:336-389): the walk of search_on_level steered by the LINKS scorer
 * (the quantized "link vectors" stored next to the links), while every candidate that is popped - the one that ends the loop included -
 * is scored by the BASE scorer (the full "base vector" stored in front of its links) into a second SearchContext, which is the result. */
static qo_topk *search_on_level_with_vectors(const qo_hnsw *g, qscore *q, const qo_scorer *base, qo_scored_point level_entry, uint32_t level,
                                             uint32_t ef, visited_t *vis, uint64_t *n_base) {
    visited_next(vis);
    visited_check_update(vis, level_entry.idx);
    qo_topk *links_nearest = qo_topk_new(ef), *base_nearest = qo_topk_new(ef);
    maxheap cands = {0, 0, 0}, base_cands = {0, 0, 0};
    process_candidate(links_nearest, &cands, level_entry);
    const uint32_t limit = level_m(g, level);
    uint32_t *ids = (uint32_t *)malloc(sizeof(uint32_t) * (g->m0 + g->m + 1));
    qo_scored_point cand;
    while (mh_pop(&cands, &cand)) {
        qo_scored_point worst;
        const float lower_bound = qo_topk_top(links_nearest, &worst) ? worst.score : -3.40282347e+38f;
        qo_scored_point b = {cand.idx, qo_scorer_score_point(base, cand.idx)};      /* base_scorer.score_bytes(base_vector) */
        (*n_base)++;
        if (cand.score < lower_bound) {
            process_candidate(base_nearest, &base_cands, b);
            break;
        }
        uint32_t n = read_links(g, cand.idx, level, ids, 0), k = 0;
        for (uint32_t i = 0; i < n; i++) if (!visited_check(vis, ids[i])) ids[k++] = ids[i];
        process_candidate(base_nearest, &base_cands, b);
        k = filter_truncate(q, ids, k, limit);
        float scores[512];
        for (uint32_t i = 0; i < k; i++) scores[i] = qs_score(q, ids[i]);
        for (uint32_t i = 0; i < k; i++) {
            qo_scored_point sp = {ids[i], scores[i]};
            process_candidate(links_nearest, &cands, sp);
            visited_check_update(vis, ids[i]);
        }
    }
    free(ids);
    free(cands.d);
    free(base_cands.d);
    qo_topk_free(links_nearest);
    return base_nearest;
}

/* GraphLayers::search_with_vectors (graph_layers.rs:564-596).  search_entry_with_vectors (:391-449) is search_entry with the links scorer
 * (score_point of the entry, then score_points(links, limit) per level): the same function here. */
uint32_t qo_hnsw_search_with_vectors(const qo_hnsw *g, const qo_scorer *links_scorer, const qo_scorer *base_scorer, uint32_t top, uint32_t ef,
                                     qo_scored_point *out, uint64_t *n_scored, uint64_t *n_base_scored) {
    qscore q = {links_scorer, NULL, 0, 0};
    uint32_t ep_id, ep_level;
    uint64_t n_base = 0;
    if (!get_entry_point(g, &q, &ep_id, &ep_level)) { if (n_scored) *n_scored = 0; if (n_base_scored) *n_base_scored = 0; return 0; }
    qo_scored_point zero_level_entry = search_entry(g, &q, ep_id, ep_level, 0, 0);
    if (ef < top) ef = top;
    visited_t vis;
    visited_init(&vis, g->n);
    qo_topk *nearest = search_on_level_with_vectors(g, &q, base_scorer, zero_level_entry, 0, ef, &vis, &n_base);
    free(vis.cnt);
    qo_scored_point *sorted = (qo_scored_point *)malloc(sizeof(qo_scored_point) * (ef + 1));
    uint32_t n = (uint32_t)qo_topk_into_sorted(nearest, sorted);
    qo_topk_free(nearest);
    if (n > top) n = top;
    memcpy(out, sorted, sizeof(qo_scored_point) * n);
    free(sorted);
    if (n_scored) *n_scored = q.n_scored;
    if (n_base_scored) *n_base_scored = n_base;
    return n;
}

static uint32_t hnsw_search_impl(const qo_hnsw *g, const qo_scorer *scorer, uint32_t top, uint32_t ef, int algorithm, qo_scored_point *out,
                                 uint64_t *n_scored, qo_scored_point *pops, uint32_t pop_cap, uint32_t *n_pops) {
    qscore q = {scorer, NULL, 0, 0, pops, pop_cap, 0};
    if (n_pops) *n_pops = 0;
    uint32_t ep_id, ep_level;
    if (!get_entry_point(g, &q, &ep_id, &ep_level)) { if (n_scored) *n_scored = 0; return 0; }
    qo_scored_point zero_level_entry = search_entry(g, &q, ep_id, ep_level, 0, 0);
    if (ef < top) ef = top;
    visited_t vis, vis2;
    visited_init(&vis, g->n);
    visited_init(&vis2, g->n);
    qo_topk *nearest = algorithm == 1 ? search_on_level_acorn(g, &q, zero_level_entry, 0, ef, &vis, &vis2)
                                      : search_on_level(g, &q, zero_level_entry, 0, ef, &vis, 0);
    free(vis.cnt);
    free(vis2.cnt);
    qo_scored_point *sorted = (qo_scored_point *)malloc(sizeof(qo_scored_point) * (ef + 1));
    uint32_t n = (uint32_t)qo_topk_into_sorted(nearest, sorted);
    qo_topk_free(nearest);
    if (n > top) n = top;
    memcpy(out, sorted, sizeof(qo_scored_point) * n);
    free(sorted);
    if (n_scored) *n_scored = q.n_scored;
    if (n_pops) *n_pops = q.trace_n;
    return n;
}
uint32_t qo_hnsw_search_algo(const qo_hnsw *g, const qo_scorer *scorer, uint32_t top, uint32_t ef, int algorithm, qo_scored_point *out,
                             uint64_t *n_scored) {
    return hnsw_search_impl(g, scorer, top, ef, algorithm, out, n_scored, NULL, 0, NULL);
}
/* GraphLayers::search with SearchAlgorithm::Hnsw + the pop sequence of its level-0 loop (the candidates that pass the lower-bound test, in order):
 * what qmx_hnsw_search_traced lists on the device.  *n_pops may exceed pop_cap (the list is then incomplete). */
uint32_t qo_hnsw_search_traced(const qo_hnsw *g, const qo_scorer *scorer, uint32_t top, uint32_t ef, qo_scored_point *out, uint64_t *n_scored,
                               qo_scored_point *pops, uint32_t pop_cap, uint32_t *n_pops) {
    return hnsw_search_impl(g, scorer, top, ef, 0, out, n_scored, pops, pop_cap, n_pops);
}

// custom_query.hip — the custom-query scorers of qdrant on top of the per-example similarity matrix.
//
// Reference: CustomQueryScorer (lib/segment/src/vector_storage/query_scorer/custom_query_scorer.rs:44-121):
//   score(point) = query.score_by(|example| Metric::similarity(example, point))
// with the Query implementations
//   RecoBestScoreQuery::score_by   vector_storage/query/reco_query.rs:68-92    (max by total_cmp, scaled_fast_sigmoid)
//   RecoSumScoresQuery::score_by   reco_query.rs:114-131                        (sequential f32 sums, pos - neg)
//   DiscoverQuery::score_by        discover_query.rs:45-73 (+ ContextPair::rank_by context_query.rs:38-45)
//   ContextQuery::score_by         context_query.rs:53-62, 112-118              (sum of fast_sigmoid(min(pos - neg - EPSILON, 0)))
//   FeedbackQuery::score_by        feedback_query.rs:198-226                    (a * sim(target) + sum pc_i * (sim(pos_i) - sim(neg_i)))
//   fast_sigmoid / scaled_fast_sigmoid  lib/common/common/src/math.rs:7-18
// The similarities are the scan kernels' (bit-identical to the x86 leaves), the combination is a handful of f32
// operations in the reference's order: bit-exact end to end.  Example order inside a query = the reference's
// flat_iter(): reco: positives, then negatives; discover: target, then (positive, negative) per pair; context: pairs.
#include "custom_combine.hpp"
#include "kernels.hpp"

namespace qmx {

// sims[e * stride] = similarity of example (first + e) with this candidate
__device__ __forceinline__ float custom_combine(const qmx_custom_query &q, const float *sims, uint64_t stride, const float *coefs) {
    const float *s = sims + (uint64_t)q.first * stride;
    return custom_score_by(q.kind, q.n_a, q.n_b, coefs + q.coef_first, [&](uint32_t e) { return s[(uint64_t)e * stride]; });
}

__global__ __launch_bounds__(256) void custom_combine_kernel(const qmx_custom_query *queries, uint32_t n_queries, const float *sims, uint64_t n,
                                                             const float *coefs, float *out) {
    const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t qi = blockIdx.y;
    if (c >= n || qi >= n_queries) return;
    out[(uint64_t)qi * n + c] = custom_combine(queries[qi], sims + c, n, coefs);
}

int32_t launch_custom_combine(hipStream_t st, const qmx_custom_query *d_queries, uint32_t n_queries, const float *d_sims, uint64_t n,
                              const float *d_coefs, float *d_out) {
    if (n == 0 || n_queries == 0) return QMX_OK;
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(custom_combine_kernel, dim3((uint32_t)((n + 255) / 256), n_queries), dim3(256), 0, st, d_queries, n_queries, d_sims, n, d_coefs, d_out);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

// Colbert MaxSim over multi-dense vectors (score_max_similarity, lib/segment/src/vector_storage/query_scorer/mod.rs:70-97):
// sims = [inner queries][inner rows] similarities (the dense scan's bits); multi-query j = inner queries [qfirst[j], qfirst[j + 1]);
// point p = inner rows [offsets[p], offsets[p + 1]).  sum over the query's inner vectors (in order, from 0.0) of the max over the point's
// inner vectors (`if sim > max_sim`, from -inf): the reference's two loops, one thread per (multi-query, candidate point).
__global__ __launch_bounds__(256) void maxsim_kernel(const float *sims, uint64_t n_rows, const uint32_t *qfirst, uint32_t n_queries,
                                                     const uint64_t *offsets, uint32_t n_points, const uint32_t *ids, uint64_t n, float *out, int *err_flag) {
    const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t j = blockIdx.y;
    if (c >= n || j >= n_queries) return;
    const uint32_t p = ids ? ids[c] : (uint32_t)c;
    if (p >= n_points) {
        *err_flag = 1;
        return;
    }
    const uint64_t b0 = offsets[p], b1 = offsets[p + 1];
    float sum = 0.0f;
    for (uint32_t a = qfirst[j]; a < qfirst[j + 1]; ++a) {
        const float *row = sims + (uint64_t)a * n_rows;
        float max_sim = -__builtin_inff();
        for (uint64_t b = b0; b < b1; ++b) {
            const float sim = row[b];
            if (sim > max_sim) max_sim = sim;
        }
        sum += max_sim;
    }
    out[(uint64_t)j * n + c] = sum;
}
int32_t launch_maxsim(hipStream_t st, const float *d_sims, uint64_t n_rows, const uint32_t *d_qfirst, uint32_t n_queries, const uint64_t *d_offsets,
                      uint32_t n_points, const uint32_t *d_ids, uint64_t n, float *d_out, int *err_flag) {
    if (n == 0 || n_queries == 0) return QMX_OK;
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(maxsim_kernel, dim3((uint32_t)((n + 255) / 256), n_queries), dim3(256), 0, st, d_sims, n_rows, d_qfirst, n_queries, d_offsets,
                       n_points, d_ids, n, d_out, err_flag);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

// top-k of a score row per query over the candidate stream (ids or 0..n), deleted / filtered points skipped:
// FixedLengthPriorityQueue + into_sorted_vec as everywhere else; top > 64 in bounded passes of 64.
// DeletedView::live for U ids at once, as masks (all ones = live): the bitmap words of the U ids are loaded before any of them is looked at (the
// branches around the loads are wave-uniform: which bitmaps exist), so the round trips overlap
template <int U>
__device__ __forceinline__ void live_masks(const DeletedView &d, const uint32_t (&id)[U], uint64_t (&keep)[U]) {
    uint64_t wp[U], wv[U], wa[U];
#pragma unroll
    for (int u = 0; u < U; ++u) wp[u] = wv[u] = wa[u] = 0;
    if (d.point_deleted && d.n_point_bits) {
#pragma unroll
        for (int u = 0; u < U; ++u) wp[u] = d.point_deleted[(id[u] < d.n_point_bits ? id[u] : 0u) >> 6];
    }
    if (d.vec_deleted && d.n_vec_bits) {
#pragma unroll
        for (int u = 0; u < U; ++u) wv[u] = d.vec_deleted[(id[u] < d.n_vec_bits ? id[u] : 0u) >> 6];
    }
    if (d.allowed && d.n_allowed_bits) {
#pragma unroll
        for (int u = 0; u < U; ++u) wa[u] = d.allowed[(id[u] < d.n_allowed_bits ? id[u] : 0u) >> 6];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint64_t bit = 1ull << (id[u] & 63u);
        const bool vdel = d.vec_deleted && id[u] < d.n_vec_bits && (wv[u] & bit);
        const bool pdel = d.point_deleted ? (id[u] < d.n_point_bits ? (wp[u] & bit) != 0 : true) : !(id[u] < d.n_rows);
        const bool ok = d.allowed ? (id[u] < d.n_allowed_bits && (wa[u] & bit)) : true;
        keep[u] = (!vdel && !pdel && ok) ? ~0ull : 0ull;
    }
}

constexpr int CT_BLOCK = 1024;
constexpr int CT_NW = CT_BLOCK / WAVE;
__global__ __launch_bounds__(CT_BLOCK) void custom_topk_kernel(const float *scores, uint64_t n, const uint32_t *ids, DeletedView del, uint32_t top,
                                                               qmx_scored_point *out, uint32_t *out_counts, uint64_t *bound_out) {
    __shared__ uint64_t sh[CT_NW][WAVE];
    __shared__ uint64_t sh_bound;
    const uint32_t q = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *row = scores + (uint64_t)q * n;
    uint64_t bound = ~0ull;
    uint32_t total = 0;
    for (uint32_t off = 0; off < top; off += WAVE) {
        const int ptop = (int)(top - off < (uint32_t)WAVE ? top - off : (uint32_t)WAVE);
        uint64_t list = 0;
        // CT_U chunks per trip: their id / score / deleted-bit loads are in flight together (the walk over a short score row - the sample
        // pre-scans hand over ~8 k scores per query - is three dependent loads per chunk otherwise)
        constexpr int CT_U = 4;
        for (uint64_t base = (uint64_t)wave * WAVE; base < n; base += (uint64_t)CT_BLOCK * CT_U) {
            uint64_t key[CT_U];
            uint32_t id[CT_U];
#pragma unroll
            for (int u = 0; u < CT_U; ++u) {
                const uint64_t c = base + (uint64_t)u * CT_BLOCK + lane;
                const uint64_t cc = c < n ? c : 0;
                id[u] = ids ? ids[cc] : (uint32_t)cc;
                key[u] = c < n ? make_key(row[cc], id[u]) : 0ull;
            }
            // (as a mask, not as `if (... || !live) key = 0`: clang 19's AMDGPU control-flow lowering dropped the assignment of that form here -
            // deleted rows came back in the lists; tests/test_gpu_custom_queries.py and the all-deleted-sample test of the split scan caught it)
            uint64_t keep[CT_U];
            live_masks<CT_U>(del, id, keep);
#pragma unroll
            for (int u = 0; u < CT_U; ++u) key[u] = key[u] < bound ? key[u] & keep[u] : 0ull;
#pragma unroll
            for (int u = 0; u < CT_U; ++u) {
                if (key[u] <= readlane_u64(list, ptop - 1)) key[u] = 0;
                uint64_t m = __ballot(key[u] != 0);
                while (m) {
                    const int src = __builtin_ctzll(m);
                    m &= m - 1;
                    const uint64_t nk = readlane_u64(key[u], src);
                    if (nk > readlane_u64(list, ptop - 1)) wave_list_insert(list, nk, lane);
                }
            }
        }
        __syncthreads();
        sh[wave][lane] = list;
        __syncthreads();
        if (wave == 0) {
            uint64_t merged = sh[0][lane];
            for (int w = 1; w < CT_NW; ++w) {
                const uint64_t key = sh[w][lane];
                uint64_t m = __ballot(key > readlane_u64(merged, ptop - 1));
                while (m) {
                    const int src = __builtin_ctzll(m);
                    m &= m - 1;
                    const uint64_t nk = readlane_u64(key, src);
                    if (nk > readlane_u64(merged, ptop - 1)) wave_list_insert(merged, nk, lane);
                }
            }
            const bool ok = lane < ptop && merged != 0;
            if (lane < ptop) {
                qmx_scored_point p;
                p.idx = ok ? key_idx(merged) : 0u;
                p.score = ok ? key_score(merged) : 0.0f;
                out[(uint64_t)q * top + off + lane] = p;
            }
            const uint32_t cnt = (uint32_t)__popcll(__ballot(ok));
            if (lane == 0) {
                sh_bound = cnt == (uint32_t)ptop ? readlane_u64(merged, ptop - 1) : 0ull;
                total += cnt;
            }
            total = (uint32_t)__builtin_amdgcn_readfirstlane((int)total);
        }
        __syncthreads();
        bound = sh_bound;
        if (bound == 0) {
            for (uint32_t i = off + WAVE + threadIdx.x; i < top; i += CT_BLOCK) out[(uint64_t)q * top + i] = qmx_scored_point{0u, 0.0f};
            break;
        }
    }
    if (threadIdx.x == 0) {
        out_counts[q] = total;
        // the k-th best key when there are k results (the key of out[top - 1]): the starting threshold of a scan
        if (bound_out) bound_out[q] = total == top ? bound : 0ull;
    }
}

// The same selection for SHORT score rows (n <= 16 384: the sample pre-scans, filtered candidate lists) without a single serial insertion: every thread
// keeps its <= 16 keys in registers; per pass of <= 64 results the wave-wide k-th largest of the 64 lane maxima is a lower bound of the k-th best key (k lanes
// hold a key that large), the largest such bound over the 16 waves prunes the row to a few times k survivors in LDS, and those are ranked against each other
// (rank = number of larger keys: keys are distinct) - rank r goes to slot r.  Same lists, same tie order (the key carries the id) as the kernel above.
constexpr int CTS_E = 16;
constexpr uint64_t CTS_MAX_N = (uint64_t)CT_BLOCK * CTS_E;
__global__ __launch_bounds__(CT_BLOCK) void custom_topk_small_kernel(const float *scores, uint32_t n, const uint32_t *ids, DeletedView del, uint32_t top,
                                                                     qmx_scored_point *out, uint32_t *out_counts, uint64_t *bound_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_cts[];
    uint64_t *surv = reinterpret_cast<uint64_t *>(smem_cts);          // [n]
    __shared__ uint64_t sh_t[CT_NW];
    __shared__ uint64_t sh_bound;
    __shared__ uint32_t sh_cnt;
    const uint32_t q = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *row = scores + (uint64_t)q * n;
    uint64_t kreg[CTS_E];
    // two round trips per 8 192 scores: the ids and scores of eight keys per thread go out together, then their deleted / filter words
#pragma unroll
    for (int e0 = 0; e0 < CTS_E; e0 += 8) {
        if ((uint32_t)e0 * CT_BLOCK >= n) {                 // (uniform: a row of up to 8 192 scores has no second half)
#pragma unroll
            for (int u = 0; u < 8; ++u) kreg[e0 + u] = 0ull;
            continue;
        }
        uint32_t id[8];
        uint64_t key[8], keep[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t c = (uint32_t)(e0 + u) * CT_BLOCK + threadIdx.x;
            const uint32_t cc = c < n ? c : 0;
            id[u] = ids ? ids[cc] : cc;
            key[u] = c < n ? make_key(row[cc], id[u]) : 0ull;
        }
        live_masks<8>(del, id, keep);
#pragma unroll
        for (int u = 0; u < 8; ++u) kreg[e0 + u] = key[u] & keep[u];
    }
    uint64_t bound = ~0ull;
    uint32_t total = 0;
    for (uint32_t off = 0; off < top; off += WAVE) {
        const uint32_t ptop = top - off < (uint32_t)WAVE ? top - off : (uint32_t)WAVE;
        uint64_t m = 0;
#pragma unroll
        for (int e = 0; e < CTS_E; ++e) {
            const uint64_t k = kreg[e] < bound ? kreg[e] : 0ull;
            m = k > m ? k : m;
        }
        uint32_t rank = 0;
        for (int j = 0; j < WAVE; ++j) rank += readlane_u64(m, j) > m ? 1u : 0u;
        const uint64_t sel = __ballot(rank == ptop - 1 && m != 0);
        const uint64_t tw = sel ? readlane_u64(m, __builtin_ctzll(sel)) : 0ull;
        if (lane == 0) sh_t[wave] = tw;
        if (threadIdx.x == 0) { sh_cnt = 0; sh_bound = 0; }
        __syncthreads();
        uint64_t t = 0;
#pragma unroll
        for (int w = 0; w < CT_NW; ++w) t = sh_t[w] > t ? sh_t[w] : t;
        // the survivors: every live key below the previous passes' bound and not below t (t = 0: fewer than ptop lanes of any wave hold a key - all of them)
#pragma unroll
        for (int e = 0; e < CTS_E; ++e) {
            const uint64_t k = kreg[e] < bound ? kreg[e] : 0ull;
            if (k != 0 && k >= t) surv[atomicAdd(&sh_cnt, 1u)] = k;
        }
        __syncthreads();
        const uint32_t cnt = sh_cnt;
        for (uint32_t i = threadIdx.x; i < cnt; i += CT_BLOCK) {
            const uint64_t my = surv[i];
            uint32_t r = 0;
            for (uint32_t j = 0; j < cnt; ++j) r += (surv[j] > my || (surv[j] == my && j < i)) ? 1u : 0u;    // (equal keys - a candidate listed twice - keep distinct slots)
            if (r < ptop) {
                qmx_scored_point p;
                p.idx = key_idx(my);
                p.score = key_score(my);
                out[(uint64_t)q * top + off + r] = p;
                if (r == ptop - 1) sh_bound = my;
            }
        }
        const uint32_t found = cnt < ptop ? cnt : ptop;
        for (uint32_t i = found + threadIdx.x; i < ptop; i += CT_BLOCK) out[(uint64_t)q * top + off + i] = qmx_scored_point{0u, 0.0f};
        total += found;
        __syncthreads();
        bound = sh_bound;                                   // the pass's k-th key when it is full, else 0: nothing is left
        if (bound == 0) {
            for (uint32_t i = off + WAVE + threadIdx.x; i < top; i += CT_BLOCK) out[(uint64_t)q * top + i] = qmx_scored_point{0u, 0.0f};
            break;
        }
        __syncthreads();                                    // (sh_bound / sh_cnt are reset at the top of the next pass)
    }
    if (threadIdx.x == 0) {
        out_counts[q] = total;
        if (bound_out) bound_out[q] = total == top ? bound : 0ull;
    }
}

int32_t launch_custom_topk(hipStream_t st, const float *d_scores, uint64_t n, const uint32_t *d_ids, const DeletedView &del, uint32_t n_queries,
                           uint32_t top, qmx_scored_point *d_out, uint32_t *d_counts, uint64_t *d_bound) {
    if (n_queries == 0) return QMX_OK;
    ::qmx::clear_stale_error();
    if (n >= 1 && n <= CTS_MAX_N && !option(OPT_NO_TOPK_SMALL)) {
        static thread_local DeviceOnce attr_once;
        if (attr_once.need()) {
            QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(custom_topk_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(CTS_MAX_N * 8)));
            attr_once.mark();
        }
        hipLaunchKernelGGL(custom_topk_small_kernel, dim3(n_queries), dim3(CT_BLOCK), (size_t)n * 8, st, d_scores, (uint32_t)n, d_ids, del, top, d_out, d_counts, d_bound);
        QMX_HIP(hipGetLastError());
        return QMX_OK;
    }
    hipLaunchKernelGGL(custom_topk_kernel, dim3(n_queries), dim3(CT_BLOCK), 0, st, d_scores, n, d_ids, del, top, d_out, d_counts, d_bound);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

}  // namespace qmx

// topk_merge.hip — bounded-queue merges on device.
//
//  merge_keys   : per-block key lists of one scan  -> final `top` per query, sorted descending
//                 (`into_sorted_vec`, lib/common/common/src/fixed_length_priority_queue.rs:63-65)
//  merge_points : per-segment / per-GPU ScoredPointOffset lists -> one list per query
//                 (`BatchResultAggregator`, lib/shard/src/search_result_aggregator.rs:50-121, for
//                 disjoint id spaces)
//  sort_scored  : rescoring tail of `postprocess_search_result`
//                 (lib/segment/src/index/vector_index_search_common.rs:73-90): sort desc, truncate
#include "kernels.hpp"

namespace qmx {

constexpr int MERGE_BLOCK = 256;
constexpr int MERGE_NW = MERGE_BLOCK / WAVE;

__device__ __forceinline__ void wave_offer(uint64_t &list, uint64_t key, int top, int lane) {
    uint64_t m = __ballot(key > readlane_u64(list, top - 1));
    while (m) {
        const int src = __builtin_ctzll(m);
        m &= m - 1;
        const uint64_t nk = readlane_u64(key, src);
        if (nk > readlane_u64(list, top - 1)) wave_list_insert(list, nk, lane);
    }
}

// block-level finish of one pass: merge the MERGE_NW wave lists, write ScoredPointOffset entries
// out[q * out_stride + out_offset .. + top), add to / set the count, and return (to every thread) the key bound
// of the next pass: the last key written, or 0 when fewer than `top` were found (nothing is left).
__device__ __forceinline__ uint64_t block_finish(uint64_t list, int top, uint32_t q, qmx_scored_point *out, uint32_t out_stride,
                                                 uint32_t out_offset, uint32_t *out_counts) {
    __shared__ uint64_t sh[MERGE_NW][WAVE];
    __shared__ uint64_t sh_bound;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();   // a previous pass may still be reading sh / sh_bound
    sh[wave][lane] = list;
    __syncthreads();
    if (wave == 0) {
        uint64_t merged = sh[0][lane];
        for (int w = 1; w < MERGE_NW; ++w) wave_offer(merged, sh[w][lane], top, lane);
        const bool ok = lane < top && merged != 0;
        if (lane < top) {
            qmx_scored_point p;
            p.idx = ok ? key_idx(merged) : 0u;
            p.score = ok ? key_score(merged) : 0.0f;
            out[(uint64_t)q * out_stride + out_offset + lane] = p;
        }
        const uint32_t cnt = (uint32_t)__popcll(__ballot(ok));
        const uint64_t last = readlane_u64(merged, top - 1);
        if (lane == 0) {
            out_counts[q] = (out_offset ? out_counts[q] : 0u) + cnt;
            sh_bound = cnt == (uint32_t)top ? last : 0ull;
        }
    }
    __syncthreads();
    return sh_bound;
}

__global__ __launch_bounds__(MERGE_BLOCK) void merge_keys_kernel(const uint64_t *partial, uint32_t n_lists,
                                                                 uint32_t qt_stride, uint32_t top,
                                                                 qmx_scored_point *out, uint32_t *out_counts, uint32_t out_stride,
                                                                 uint32_t out_offset, uint64_t *next_bound, const int *run_if, const uint32_t *out_map,
                                                                 uint32_t shared_grid) {
    if (run_if && *run_if == 0) return;      // the exact pass behind the split prefilter was not needed (scan_split.hip)
    const uint32_t q = blockIdx.x;
    if (shared_grid) {      // pq_scan_kernel's packed pass: its `shared_grid` blocks were divided among the *run_if listed queries (pq.hip)
        const uint32_t cnt = (uint32_t)*run_if, nq_eff = cnt < gridDim.x ? cnt : gridDim.x;
        n_lists = shared_grid / nq_eff;
        qt_stride = nq_eff;
    }
    // out_map: list q of the scan belongs to query out_map[q] of the batch (the packed exact pass behind the prefilter); 0xFFFFFFFF = a padding slot
    const uint32_t oq = out_map ? out_map[q] : q;
    if (oq == 0xFFFFFFFFu) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t list = 0;
    for (uint32_t l = wave; l < n_lists; l += MERGE_NW) {
        const uint64_t key = lane < (int)top ? partial[((uint64_t)l * qt_stride + q) * top + lane] : 0;
        wave_offer(list, key, (int)top, lane);
    }
    const uint64_t nb = block_finish(list, (int)top, oq, out, out_stride, out_offset, out_counts);
    if (next_bound && threadIdx.x == 0) next_bound[q] = nb;
}

// Items are visited in the aggregator's order (list-major, then position); an id that was already
// pushed by an earlier item is dropped whatever its score (`if self.seen.insert(point.id)`,
// search_result_aggregator.rs:28-37).  The id of every item is staged in LDS and each item looks
// for an earlier twin: O(T^2 / 256) compares per thread, T = n_lists * k is a few hundred at most.
// k > 64 runs in passes of 64: pass p keeps the best 64 keys below the last key of pass p - 1.
__global__ __launch_bounds__(MERGE_BLOCK) void merge_points_kernel(const qmx_scored_point *lists,
                                                                   const uint32_t *list_counts,
                                                                   const uint32_t *list_idx_base, uint32_t n_lists,
                                                                   uint32_t nq, uint32_t k, qmx_scored_point *out,
                                                                   uint32_t *out_counts, uint64_t list_stride, uint64_t count_stride) {
    // list l = lists + l * list_stride entries, its counts = list_counts + l * count_stride words: nq * k and nq for contiguous arrays; the packed
    // records of qmx_merge_topk_packed_async (one all-gathered buffer per rank: lists, then counts) have their own strides
    extern __shared__ uint32_t seen_ids[];   // [n_lists * k]; 0xFFFFFFFF = no item or a duplicate
    const uint32_t q = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t total = n_lists * k;
    for (uint32_t j = threadIdx.x; j < total; j += MERGE_BLOCK) {
        const uint32_t l = j / k, i = j - l * k;
        const uint32_t cnt = list_counts ? list_counts[(uint64_t)l * count_stride + q] : k;
        seen_ids[j] = i < cnt ? lists[(uint64_t)l * list_stride + (uint64_t)q * k + i].idx + (list_idx_base ? list_idx_base[l] : 0u)
                              : 0xFFFFFFFFu;
    }
    __syncthreads();
    uint64_t bound = ~0ull;
    for (uint32_t off = 0; off < k; off += WAVE) {
        const int top = (int)(k - off < (uint32_t)WAVE ? k - off : (uint32_t)WAVE);
        uint64_t list = 0;
        for (uint32_t l = wave; l < n_lists; l += MERGE_NW) {
            const uint32_t cnt = list_counts ? list_counts[(uint64_t)l * count_stride + q] : k;
            for (uint32_t base = 0; base < k; base += WAVE) {
                const uint32_t i = base + lane;
                uint64_t key = 0;
                if (i < k && i < cnt) {
                    const uint32_t j = l * k + i;
                    const uint32_t id = seen_ids[j];
                    bool dup = false;
                    for (uint32_t e = 0; e < j; ++e) dup = dup || (seen_ids[e] == id);
                    if (!dup) key = make_key(lists[(uint64_t)l * list_stride + (uint64_t)q * k + i].score, id);
                    if (key >= bound) key = 0;
                }
                wave_offer(list, key, top, lane);
            }
        }
        bound = block_finish(list, top, q, out, k, off, out_counts);
        if (bound == 0) {   // exhausted: clear the tail of the row
            for (uint32_t i = off + WAVE + threadIdx.x; i < k; i += MERGE_BLOCK) out[(uint64_t)q * k + i] = qmx_scored_point{0u, 0.0f};
            break;
        }
    }
}

__global__ __launch_bounds__(MERGE_BLOCK) void sort_scored_kernel(const float *scores, const uint32_t *ids,
                                                                  const uint32_t *counts, uint32_t n_per_query,
                                                                  uint32_t top, qmx_scored_point *out,
                                                                  uint32_t *out_counts, const uint32_t *offsets) {
    const uint32_t q = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // ragged lists (offsets): query q's entries are [offsets[q], offsets[q] + counts[q]); else slots of n_per_query entries, counts[q] of them live
    const uint32_t cnt = offsets ? counts[q] : counts ? (counts[q] < n_per_query ? counts[q] : n_per_query) : n_per_query;
    const uint64_t first = offsets ? (uint64_t)offsets[q] : (uint64_t)q * n_per_query;
    uint64_t bound = ~0ull;
    for (uint32_t off = 0; off < top; off += WAVE) {
        const int ptop = (int)(top - off < (uint32_t)WAVE ? top - off : (uint32_t)WAVE);
        uint64_t list = 0;
        for (uint32_t base = wave * WAVE; base < cnt; base += MERGE_BLOCK) {
            const uint32_t i = base + lane;
            uint64_t key = i < cnt ? make_key(scores[first + i], ids[first + i]) : 0;
            if (key >= bound) key = 0;
            wave_offer(list, key, ptop, lane);
        }
        bound = block_finish(list, ptop, q, out, top, off, out_counts);
        if (bound == 0) {
            for (uint32_t i = off + WAVE + threadIdx.x; i < top; i += MERGE_BLOCK) out[(uint64_t)q * top + i] = qmx_scored_point{0u, 0.0f};
            break;
        }
    }
}

int32_t launch_merge_keys(hipStream_t st, const uint64_t *partial, uint32_t n_lists, uint32_t qt_stride,
                          uint32_t nq, uint32_t top, qmx_scored_point *out, uint32_t *out_counts, uint32_t out_stride,
                          uint32_t out_offset, uint64_t *next_bound, const int *run_if, const uint32_t *out_map, uint32_t shared_grid) {
    if (nq == 0) return QMX_OK;
    if (out_stride == 0) out_stride = top;
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(merge_keys_kernel, dim3(nq), dim3(MERGE_BLOCK), 0, st, partial, n_lists, qt_stride, top, out, out_counts,
                       out_stride, out_offset, next_bound, run_if, out_map, shared_grid);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}
int32_t launch_merge_points(hipStream_t st, const qmx_scored_point *lists, const uint32_t *list_counts,
                            const uint32_t *list_idx_base, uint32_t n_lists, uint32_t nq, uint32_t k,
                            qmx_scored_point *out, uint32_t *out_counts, uint64_t list_stride, uint64_t count_stride) {
    if (nq == 0) return QMX_OK;
    if (list_stride == 0) list_stride = (uint64_t)nq * k;
    if (count_stride == 0) count_stride = nq;
    QMX_REQUIRE((uint64_t)n_lists * k * sizeof(uint32_t) <= 64 * 1024, QMX_ERR_NOT_SUPPORTED,
                "merge of %u lists x %u entries exceeds the 64 KiB seen-id table", n_lists, k);
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(merge_points_kernel, dim3(nq), dim3(MERGE_BLOCK), (size_t)n_lists * k * sizeof(uint32_t), st, lists, list_counts, list_idx_base, n_lists, nq, k, out, out_counts,
                       list_stride, count_stride);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}
int32_t launch_sort_scored(hipStream_t st, const float *scores, const uint32_t *ids, const uint32_t *counts,
                           uint32_t n_per_query, uint32_t nq, uint32_t top, qmx_scored_point *out, uint32_t *out_counts, const uint32_t *offsets) {
    if (nq == 0) return QMX_OK;
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(sort_scored_kernel, dim3(nq), dim3(MERGE_BLOCK), 0, st, scores, ids, counts, n_per_query, top, out, out_counts, offsets);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

// packed level-0 link table of an HNSW graph: row p = [count, links...] (hnsw.hpp reads it with one round trip per hop)
__global__ void hnsw_pack_level0_kernel(const uint64_t *offsets, const uint32_t *neighbors, uint32_t n_points, uint32_t stride, uint32_t *l0) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t p = gid / stride;
    const uint32_t slot = (uint32_t)(gid % stride);
    if (p >= n_points) return;
    const uint64_t o0 = offsets[p], len = offsets[p + 1] - o0;
    l0[gid] = slot == 0 ? (uint32_t)len : (slot - 1 < len ? neighbors[o0 + slot - 1] : 0u);
}
int32_t launch_hnsw_pack_level0(hipStream_t st, const uint64_t *offsets, const uint32_t *neighbors, uint32_t n_points, uint32_t stride, uint32_t *l0) {
    if (n_points == 0) return QMX_OK;
    const uint64_t total = (uint64_t)n_points * stride;
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(hnsw_pack_level0_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, offsets, neighbors, n_points, stride, l0);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

// row p of l0x = [the m0 = stride - 1 link slots of row p of l0, 0xFFFFFFFF behind the last link][aux of every link: f32 bits] (2 m0 dwords: no count word)
__global__ void hnsw_pack_level0_aux_kernel(const uint32_t *l0, uint32_t n_points, uint32_t stride, const float *aux, uint64_t n_aux, uint32_t *l0x) {
    const uint32_t m0 = stride - 1, stride_x = 2 * m0;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t p = gid / m0;
    const uint32_t slot = (uint32_t)(gid % m0);
    if (p >= n_points) return;
    const uint32_t cnt = l0[p * stride];
    const uint32_t v = l0[p * stride + 1 + slot];
    const bool on = slot < cnt;
    l0x[p * stride_x + slot] = on ? v : 0xFFFFFFFFu;
    l0x[p * stride_x + m0 + slot] = (on && v < n_aux) ? __float_as_uint(aux[v]) : 0u;
}
int32_t launch_hnsw_pack_level0_aux(hipStream_t st, const uint32_t *l0, uint32_t n_points, uint32_t stride, const float *aux, uint64_t n_aux, uint32_t *l0x) {
    if (n_points == 0) return QMX_OK;
    const uint64_t total = (uint64_t)n_points * (stride - 1);
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(hnsw_pack_level0_aux_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, l0, n_points, stride, aux, n_aux, l0x);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

// candidates [nq][n_per] (ScoredPointOffset) -> ids [nq][n_per] (for the rescoring pass) and, when out != nullptr, the
// first `top` entries + clamped counts (no rescoring: search_result.truncate(top), vector_index_search_common.rs:89)
__global__ void split_candidates_kernel(const qmx_scored_point *cand, const uint32_t *cand_cnt, uint32_t n_per, uint32_t nq, uint32_t *ids,
                                        uint32_t top, qmx_scored_point *out, uint32_t *out_counts) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (uint64_t)nq * n_per) return;
    const uint32_t q = (uint32_t)(gid / n_per), i = (uint32_t)(gid % n_per);
    const qmx_scored_point p = cand[gid];
    if (ids) ids[gid] = p.idx;
    if (out) {
        if (i < top) out[(uint64_t)q * top + i] = i < cand_cnt[q] ? p : qmx_scored_point{0u, 0.0f};
        if (i == 0) out_counts[q] = cand_cnt[q] < top ? cand_cnt[q] : top;
    }
}
int32_t launch_split_candidates(hipStream_t st, const qmx_scored_point *cand, const uint32_t *cand_cnt, uint32_t n_per, uint32_t nq,
                                uint32_t *ids, uint32_t top, qmx_scored_point *out, uint32_t *out_counts) {
    const uint64_t total = (uint64_t)nq * n_per;
    if (total == 0) return QMX_OK;
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL(split_candidates_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, cand, cand_cnt, n_per, nq, ids, top, out, out_counts);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

}  // namespace qmx

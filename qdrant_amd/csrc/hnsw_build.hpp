// hnsw_build.hpp — HNSW graph construction on the device, batch-parallel.
//
// Replaces GraphLayersBuilder::link_new_point (lib/segment/src/index/hnsw_index/graph_layers_builder.rs:417-474:
// search_entry above the point's level, then per level search_on_level(ef_construct) -> link_with_heuristic
// :532-556) and LinksContainer::{fill_from_sorted_with_heuristic :47-71, connect_with_heuristic :106-132}
// for dense f32 / f16 storages and for SQ-int8 storages (the reference builds through the quantized scorer when the segment
// has one, hnsw/build.rs:334-341: FilteredScorer::new_internal over QuantizedVectors).  The reference builds with a rayon pool of threads inserting points concurrently
// under per-point locks (hnsw/build.rs:355) and has a Vulkan batch builder (hnsw_index/gpu/*); neither has a
// deterministic insertion order, so — like for those — parity of a built graph is defined by its invariants and
// by search quality (recall equal to the CPU oracle's build within noise), not link-for-link.
//
// Scheme: points are inserted in id order, in batches that never exceed 1/32 of the points already in the graph.
//   phase 1 (hnsw_build_search_kernel)  one wavefront per new point, graph READ-ONLY: greedy descent to the point's
//            level, then on every level l <= level(p): beam search with ef_construct (same register beam / visited
//            bitmap as the search kernel, hnsw.hpp) -> candidates -> heuristic selection -> sel[p][l] (+ scores)
//   phase 2 (hnsw_build_link_kernel)    one wavefront per new point, no searches running: publish p's own link
//            lists, then for every selected neighbour q: lock(q), append p or — when q is full — re-select q's links
//            among links(q) + p with the same heuristic, unlock.
// Scores are the scan's lane policies (bit-identical to the x86 reference); a stored row is used as a "query entry"
// directly (dense rows need no aux block; an SQ code row takes its query offset from ScanArgs::sq_qoff, see query_args),
// the new point's row is staged in LDS.
//
// Graph storage while building: fixed-capacity lists, level 0: links0[p][m0] + cnt0[p]; levels >= 1:
// linksU[up_off[p] + l - 1][m] + cntU[...].  Exported afterwards to the plain GraphLinks arrays.
#pragma once
#include "hnsw.hpp"

namespace qmx {

// loads / stores of link lists in phase 2 go to L2 (agent scope): another CU may have rewritten the list under its lock
__device__ __forceinline__ uint32_t ld_agent(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The scan arguments for scoring against stored row `qid` AS THE QUERY (FilteredScorer::new_internal, point_scorer.rs:183-218).
// Dense rows need nothing.  An SQ row is its codes plus vector_offset; as a query its offset is vector_offset - shift
// (encode_internal_vector, encoded_vectors_u8.rs:715-728 == postprocess_internal_score :105-114): handed to the policy in sq_qoff.
template <class H>
__device__ __forceinline__ ScanArgs query_args(const ScanArgs &a, uint32_t qid) {
    ScanArgs b = a;
    if constexpr (H::INTERNAL_QOFF) b.sq_qoff = a.row_offsets[qid] - a.sq_shift;
    if constexpr (H::INTERNAL_NORM) {   // per-pair u8 cosine: the stored row's own norm is the query norm (cosine.rs computes both the same way)
        b.u8_qnorm_f = a.row_norms_f[qid];
        b.u8_qnorm_i = a.row_norms_i[qid];
    }
    if constexpr (is_maxsim_q<H>::value) b.mv_q_tokens = (uint32_t)(a.mv_offsets[qid + 1] - a.mv_offsets[qid]);      // the point's tokens among the batch's entries
    return b;
}

// The query entry of stored point `id`: its row; for multi-vector points (HopMaxSimInternal) the id itself, carried in the pointer.
template <class H>
__device__ __forceinline__ const unsigned char *stored_query(const ScanArgs &a, uint32_t id) {
    if constexpr (is_maxsim_internal<H>::value) return reinterpret_cast<const unsigned char *>((uintptr_t)id);
    else return reinterpret_cast<const unsigned char *>(a.rows) + (uint64_t)id * a.row_stride;
}

// fill_from_sorted_with_heuristic (links_container.rs:47-71): candidates sorted by descending score to the target;
// keep c unless it is closer to an already kept link than to the target.  cand_* and sel_* live in LDS.
// Returns the number kept (wave-uniform).  `a.rows` rows double as query entries.
template <class H>
__device__ __forceinline__ uint32_t heuristic_fill(const ScanArgs &a, const uint32_t *cand_ids, const float *cand_scores, uint32_t n_cand,
                                                   uint32_t lm, uint32_t *sel_ids, float *sel_scores, uint32_t *hop_ids, float *hop_scores,
                                                   int lane) {
    uint32_t n_sel = 0;
    for (uint32_t c = 0; c < n_cand && n_sel < lm; ++c) {
        const uint32_t cid = cand_ids[c];
        const float cs = cand_scores[c];
        bool skip = false;
        for (uint32_t base = 0; base < n_sel && !skip; base += 64) {
            const uint32_t k = n_sel - base < 64 ? n_sel - base : 64;
            __syncthreads();
            if ((uint32_t)lane < k) hop_ids[lane] = sel_ids[base + lane];
            hop_score<H>(query_args<H>(a, cid), stored_query<H>(a, cid), hop_ids, hop_scores, k, lane);   // score(candidate, kept link)
            const bool bad = (uint32_t)lane < k && hop_scores[lane] > cs;
            skip = __ballot(bad) != 0;
        }
        if (skip) continue;
        __syncthreads();
        if (lane == 0) {
            sel_ids[n_sel] = cid;
            sel_scores[n_sel] = cs;
        }
        ++n_sel;
    }
    __syncthreads();
    return n_sel;
}

// entries of the candidate arrays of phase 1 (LDS): the register beam's 64 E, or ef_construct rounded up to whole waves for the LDS beam
__host__ __device__ static inline uint32_t hnsw_build_cand_cap(int E, uint32_t ef) { return E ? 64u * (uint32_t)E : (ef + 63u) / 64u * 64u; }

// ---- phase 1 ----------------------------------------------------------------------------------------------------
// H scores the searches of an insertion (the new point as the query), HI scores stored <-> stored pairs (score_internal: the heuristic).
// They are the same policy except for storages without an internal query (PQ): H = the LUT of the original vector, HI = centroid tables.
template <class H, class HI, int E>
__global__ __launch_bounds__(64) void hnsw_build_search_kernel(const ScanArgs a0, const HnswBuildArgs h) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    // E > 0: the beam of the insertion searches in registers (64 E entries); E == 0 (ef_construct > 512): in LDS behind the query entry, as the walk's
    const uint32_t ccap = hnsw_build_cand_cap(E, h.ef_construct);
    uint32_t *hop_ids = reinterpret_cast<uint32_t *>(smem);
    float *hop_scores = reinterpret_cast<float *>(smem + 256);
    uint32_t *cand_ids = reinterpret_cast<uint32_t *>(smem + 512);                     // [ccap]
    float *cand_scores = reinterpret_cast<float *>(smem + 512 + 4 * (size_t)ccap);
    uint32_t *sel_ids = reinterpret_cast<uint32_t *>(smem + 512 + 8 * (size_t)ccap);   // [HNSW_BUILD_MAX_M0]
    float *sel_scores = reinterpret_cast<float *>(smem + 512 + 8 * (size_t)ccap + 4 * HNSW_BUILD_MAX_M0);
    unsigned char *q_lds = smem + 512 + 8 * (size_t)ccap + 8 * HNSW_BUILD_MAX_M0;
    unsigned char *beam_lds = q_lds + ((size_t)h.lds_query_bytes + 15) / 16 * 16;
    uint32_t *vis = h.visited + (uint64_t)blockIdx.x * h.vis_words;
    uint32_t *vlog = h.vis_log + (uint64_t)blockIdx.x * h.log_cap;
    const unsigned char *rows = reinterpret_cast<const unsigned char *>(a0.rows);
    const uint32_t ef = h.ef_construct;

    // the insertions of a batch are handed out one at a time (h.next[0], set to the grid size by the host): the static stride paid ceil(count / slots)
    // insertions of the slowest slot per batch (hnsw.hpp does the same for the searches)
    auto next_item = [&](uint32_t bi) -> uint32_t {
        if (!h.next) return bi + gridDim.x;
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(&h.next[0], 1u);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    };
    for (uint32_t bi = blockIdx.x; bi < h.count; bi = next_item(bi)) {
        const uint32_t p = h.first + bi;
        const uint32_t lp = h.level[p];
        const ScanArgs a = query_args<H>(a0, p);     // the new point is the query of every score below
        uint32_t *my_cnt = h.sel_cnt + (uint64_t)bi * HNSW_BUILD_MAX_LEVELS;
        if (lane < (int)HNSW_BUILD_MAX_LEVELS) my_cnt[lane] = 0;
        if (!a.del.live(p)) continue;            // deleted points are never indexed (hnsw/build.rs:293-300)

        // the query entry of the new point: made from its original vector before this launch (batch_queries), or its own stored row
        // staged in LDS (zero padded to whole 128-byte steps)
        const unsigned char *qp = q_lds;
        __syncthreads();
        if constexpr (is_maxsim_internal<H>::value) {       // a multi-vector point: its inner rows stay in HBM
            qp = stored_query<H>(a0, p);
        } else if (h.batch_queries && h.lds_query_bytes) {        // a small entry (TurboQuant): staged like a row (a policy may own scratch behind it)
            const unsigned char *src = h.batch_queries + (uint64_t)bi * h.batch_q_stride;
            const uint32_t q_bytes = h.lds_query_bytes < h.batch_q_stride ? h.lds_query_bytes : (uint32_t)h.batch_q_stride;
            for (uint32_t i = (uint32_t)lane * 16; i < q_bytes; i += 64 * 16)
                *reinterpret_cast<uint4 *>(q_lds + i) = *reinterpret_cast<const uint4 *>(src + i);
        } else if (h.batch_queries) {                       // a LUT (PQ): read through L2
            qp = h.batch_queries + (uint64_t)bi * h.batch_q_stride;
            // multi-vector points: one entry per inner vector of the batch, point p's from its first inner row on
            if constexpr (is_maxsim_q<H>::value) qp = h.batch_queries + (a0.mv_offsets[p] - a0.mv_offsets[h.first]) * h.batch_q_stride;
        } else {
            const unsigned char *src = rows + (uint64_t)p * a.row_stride;
            for (uint32_t i = (uint32_t)lane * 16; i < h.lds_query_bytes; i += 64 * 16) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (i + 16 <= h.row_bytes) v = *reinterpret_cast<const uint4 *>(src + i);
                else if (i < h.row_bytes) {
                    unsigned char tmp[16];
                    for (uint32_t b = 0; b < 16; ++b) tmp[b] = i + b < h.row_bytes ? src[i + b] : 0;
                    v = *reinterpret_cast<const uint4 *>(tmp);
                }
                *reinterpret_cast<uint4 *>(q_lds + i) = v;
            }
        }
        __syncthreads();

        // ---- search_entry: greedy descent from the entry point's level to level(p) + 1 ----
        uint32_t cur_id = h.ep_id;
        float cur_score;
        {
            if (lane == 0) hop_ids[0] = cur_id;
            // graph_layers_builder.rs:441-447: an entry point at or below the new point's level is taken with score_internal(p, entry)
            if (is_asymmetric<HI>::value && h.ep_level <= lp)
                hop_score<HI>(query_args<HI>(a0, p), stored_query<HI>(a0, p), hop_ids, hop_scores, 1, lane);
            else
                hop_score<H>(a, qp, hop_ids, hop_scores, 1, lane);
            cur_score = hop_scores[0];
        }
        for (uint32_t level = h.ep_level; level > lp; --level) {
            bool changed = true;
            while (changed) {
                changed = false;
                const uint32_t *lst = h.g.list(cur_id, level);
                const uint32_t len = *h.g.count(cur_id, level);
                for (uint32_t base = 0; base < len; base += 64) {
                    const uint32_t i = base + (uint32_t)lane;
                    const bool on = i < len;
                    const uint32_t id = on ? lst[i] : 0;
                    const uint64_t mask = __ballot(on);
                    const uint32_t k = (uint32_t)__popcll(mask);
                    __syncthreads();
                    if (on) hop_ids[lane] = id;
                    hop_score<H>(a, qp, hop_ids, hop_scores, k, lane);
                    uint64_t mk = 0;
                    if ((uint32_t)lane < k && hop_scores[lane] > cur_score)
                        mk = ((uint64_t)score_to_ord(hop_scores[lane]) << 32) | (uint32_t)(~(uint32_t)lane);
                    const uint64_t best = wave_max_u64(mk);
                    if (best) {
                        const uint32_t bl = ~(uint32_t)best;
                        cur_id = hop_ids[bl];
                        cur_score = hop_scores[bl];
                        changed = true;
                    }
                }
            }
        }

        // ---- per level: search_on_level(ef_construct) -> heuristic -> sel ----
        const uint32_t top_link_level = lp < h.ep_level ? lp : h.ep_level;
        for (int32_t lv = (int32_t)top_link_level; lv >= 0; --lv) {
            const uint32_t level = (uint32_t)lv;
            Beam<E> beam;
            if constexpr (E == 0) {
                __syncthreads();
                beam.init(beam_lds, ef);
            }
            beam.clear();
            uint32_t log_cnt = 1;
            if (lane == 0) {
                atomicOr(&vis[cur_id >> 5], 1u << (cur_id & 31));
                vlog[0] = cur_id >> 5;
            }
            beam.insert(make_key(cur_score, cur_id), ef, lane);
            while (true) {
                const uint64_t ck = beam.pop_best(lane);
                if (ck == 0) break;
                const uint32_t cand = key_idx(ck);
                const uint32_t *lst = h.g.list(cand, level);
                const uint32_t len = *h.g.count(cand, level);
                for (uint32_t base = 0; base < len; base += 64) {
                    const uint32_t i = base + (uint32_t)lane;
                    const bool on = i < len;
                    const uint32_t id = on ? lst[i] : 0;
                    const bool live = on && id < h.n_points;
                    const uint32_t bit = 1u << (id & 31);
                    const uint32_t old = live ? atomicOr(&vis[id >> 5], bit) : bit;
                    const bool keep = live && !(old & bit);
                    const uint64_t mask = __ballot(keep);
                    const uint32_t rank = (uint32_t)__popcll(mask & lt_mask);
                    const uint32_t k = (uint32_t)__popcll(mask);
                    __syncthreads();
                    if (keep) {
                        hop_ids[rank] = id;
                        if (log_cnt + rank < h.log_cap) vlog[log_cnt + rank] = id >> 5;
                    }
                    log_cnt += k;
                    const uint32_t ks = k;
                    hop_score<H>(a, qp, hop_ids, hop_scores, ks, lane);
                    const uint64_t mykey = (uint32_t)lane < ks ? make_key(hop_scores[lane], hop_ids[lane]) : 0;
                    uint64_t mm = __ballot(mykey > beam.at(ef - 1));
                    while (mm) {
                        const int src = __builtin_ctzll(mm);
                        mm &= mm - 1;
                        const uint64_t nk = readlane_u64(mykey, src);
                        if (nk > beam.at(ef - 1)) beam.insert(nk, ef, lane);
                    }
                }
            }
            // candidates, best first
            __syncthreads();
            uint32_t n_cand = 0;
            if constexpr (E == 0) {
                for (uint32_t base = 0; base < ef; base += 64) {
                    const uint32_t idx = base + (uint32_t)lane;
                    const uint64_t k = idx < ef ? beam.at(idx) : 0ull;
                    const bool ok = k != 0;
                    if (ok) {
                        cand_ids[idx] = key_idx(k);
                        cand_scores[idx] = key_score(k);
                    }
                    n_cand += (uint32_t)__popcll(__ballot(ok));
                }
            } else {
#pragma unroll
            for (int e = 0; e < (E ? E : 1); ++e) {
                const uint32_t idx = (uint32_t)e * 64 + (uint32_t)lane;
                const bool ok = beam.key[e] != 0;
                if (ok) {
                    cand_ids[idx] = key_idx(beam.key[e]);
                    cand_scores[idx] = key_score(beam.key[e]);
                }
                n_cand += (uint32_t)__popcll(__ballot(ok));
            }
            }
            __syncthreads();
            // give the visited bitmap back all-zero (fresh VisitedList per level, graph_layers.rs:108-116)
            if (log_cnt <= h.log_cap) {
                for (uint32_t i = (uint32_t)lane; i < log_cnt; i += 64) vis[vlog[i]] = 0;
            } else {
                for (uint64_t w = (uint64_t)lane; w < h.vis_words; w += 64) vis[w] = 0;
            }
            __threadfence();
            if (n_cand) {                       // next level starts from the nearest candidate (:512-517)
                cur_id = cand_ids[0];
                cur_score = cand_scores[0];
            }
            const uint32_t lm = h.g.level_m(level);
            const uint32_t n_sel = heuristic_fill<HI>(a0, cand_ids, cand_scores, n_cand, lm, sel_ids, sel_scores, hop_ids, hop_scores, lane);
            const uint64_t so = ((uint64_t)bi * HNSW_BUILD_MAX_LEVELS + level) * h.g.m0;
            for (uint32_t i = (uint32_t)lane; i < n_sel; i += 64) {
                h.sel_ids[so + i] = sel_ids[i];
                h.sel_scores[so + i] = sel_scores[i];
            }
            if (lane == 0) my_cnt[level] = n_sel;
            __syncthreads();
        }
    }
}

// ---- phase 2 ----------------------------------------------------------------------------------------------------
template <class H>
__global__ __launch_bounds__(64) void hnsw_build_link_kernel(const ScanArgs a, const HnswBuildArgs h) {
    __shared__ uint32_t hop_ids[64];
    __shared__ float hop_scores[64];
    __shared__ uint32_t cand_ids[HNSW_BUILD_MAX_M0 + 16];
    __shared__ float cand_scores[HNSW_BUILD_MAX_M0 + 16];
    __shared__ uint32_t srt_ids[HNSW_BUILD_MAX_M0 + 16];
    __shared__ float srt_scores[HNSW_BUILD_MAX_M0 + 16];
    __shared__ uint32_t sel_ids[HNSW_BUILD_MAX_M0];
    __shared__ float sel_scores[HNSW_BUILD_MAX_M0];
    const int lane = threadIdx.x;
    auto next_item = [&](uint32_t bi) -> uint32_t {      // (phase 2 draws from h.next[1])
        if (!h.next) return bi + gridDim.x;
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(&h.next[1], 1u);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    };
    for (uint32_t bi = blockIdx.x; bi < h.count; bi = next_item(bi)) {
        const uint32_t p = h.first + bi;
        const uint32_t lp = h.level[p];
        const uint32_t *my_cnt = h.sel_cnt + (uint64_t)bi * HNSW_BUILD_MAX_LEVELS;
        for (uint32_t level = 0; level <= lp && level < HNSW_BUILD_MAX_LEVELS; ++level) {
            const uint32_t n_sel = my_cnt[level];
            if (n_sel == 0) continue;
            const uint64_t so = ((uint64_t)bi * HNSW_BUILD_MAX_LEVELS + level) * h.g.m0;
            const uint32_t lm = h.g.level_m(level);
            // p's own list: nobody else touches it during this phase (p was not in the graph in phase 1)
            uint32_t *mine = h.g.list(p, level);
            for (uint32_t i = (uint32_t)lane; i < n_sel; i += 64) st_agent(mine + i, h.sel_ids[so + i]);
            if (lane == 0) st_agent(h.g.count(p, level), n_sel);
            // back links: LinksContainer::connect_with_heuristic(p, q) under q's lock
            for (uint32_t k = 0; k < n_sel; ++k) {
                const uint32_t q = h.sel_ids[so + k];
                float s_qp = h.sel_scores[so + k];             // score(p, q) == score(q, p), bit for bit ...
                if constexpr (is_asymmetric<H>::value) {       // ... unless the search score is not the storage's score_internal (PQ)
                    __syncthreads();
                    if (lane == 0) hop_ids[0] = p;
                    hop_score<H>(query_args<H>(a, q), stored_query<H>(a, q), hop_ids, hop_scores, 1, lane);
                    s_qp = hop_scores[0];
                }
                if (lane == 0) {
                    while (atomicCAS(&h.lock[q], 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(8);
                }
                __syncthreads();
                __threadfence();
                uint32_t *lst = h.g.list(q, level);
                uint32_t *cntp = h.g.count(q, level);
                const uint32_t len = ld_agent(cntp);
                if (len < lm) {
                    if (lane == 0) {
                        st_agent(lst + len, p);
                        st_agent(cntp, len + 1);
                    }
                } else {
                    // candidates = links(q) + p with their scores to q, sorted descending (total_cmp; equal keys keep
                    // input order, the new point last), then the heuristic again
                    const uint32_t n_c = len + 1;               // <= m0 + 1 <= HNSW_BUILD_MAX_M0 + 1
                    for (uint32_t base = 0; base < len; base += 64) {
                        const uint32_t kk = len - base < 64 ? len - base : 64;
                        __syncthreads();
                        if ((uint32_t)lane < kk) {
                            const uint32_t id = ld_agent(lst + base + lane);
                            hop_ids[lane] = id;
                            cand_ids[base + lane] = id;
                        }
                        hop_score<H>(query_args<H>(a, q), stored_query<H>(a, q), hop_ids, hop_scores, kk, lane);
                        if ((uint32_t)lane < kk) cand_scores[base + lane] = hop_scores[lane];
                    }
                    if (lane == 0) {
                        cand_ids[len] = p;
                        cand_scores[len] = s_qp;
                    }
                    __syncthreads();
                    for (uint32_t i = (uint32_t)lane; i < n_c; i += 64) {
                        const uint32_t oi = score_to_ord(cand_scores[i]);
                        uint32_t rank = 0;
                        for (uint32_t j = 0; j < n_c; ++j) {
                            const uint32_t oj = score_to_ord(cand_scores[j]);
                            rank += (oj > oi || (oj == oi && j < i)) ? 1u : 0u;
                        }
                        srt_ids[rank] = cand_ids[i];
                        srt_scores[rank] = cand_scores[i];
                    }
                    __syncthreads();
                    const uint32_t n_new = heuristic_fill<H>(a, srt_ids, srt_scores, n_c, lm, sel_ids, sel_scores, hop_ids, hop_scores, lane);
                    for (uint32_t i = (uint32_t)lane; i < n_new; i += 64) st_agent(lst + i, sel_ids[i]);
                    if (lane == 0) st_agent(cntp, n_new);
                }
                __threadfence();
                __syncthreads();
                if (lane == 0) atomicExch(&h.lock[q], 0u);
            }
        }
    }
}

// H: the scorer of the insertion searches; HI: the stored <-> stored scorer (phase 2 and the heuristic), H itself unless given
template <class H, class HI = H>
int32_t launch_hnsw_build_hop(hipStream_t st, const ScanArgs &a, const HnswBuildArgs &h, int phase, uint32_t grid, int *per_cu) {
    QMX_REQUIRE(h.ef_construct >= 1 && h.ef_construct <= HNSW_MAX_EF, QMX_ERR_NOT_SUPPORTED, "ef_construct %u not in 1..%u", h.ef_construct, HNSW_MAX_EF);
    // the beam of the insertion searches: 128 / 512 entries in registers, wider ones in LDS behind the query entry (as the walk's, hnsw.hpp Beam<0>)
    const int e_sel = h.ef_construct <= 128 ? 2 : h.ef_construct <= HNSW_MAX_EF_REG ? 8 : 0;
    const size_t lds1 = 512 + 8 * (size_t)hnsw_build_cand_cap(e_sel, h.ef_construct) + 8 * HNSW_BUILD_MAX_M0 + ((size_t)h.lds_query_bytes + 15) / 16 * 16 +
                        (e_sel == 0 ? hnsw_beam_lds(h.ef_construct) : 0);
    if (phase == 1) {
        auto k2 = hnsw_build_search_kernel<H, HI, 2>;
        auto k8 = hnsw_build_search_kernel<H, HI, 8>;
        auto k0 = hnsw_build_search_kernel<H, HI, 0>;
        QMX_REQUIRE(lds1 <= 160 * 1024, QMX_ERR_NOT_SUPPORTED, "HNSW build: ef_construct %u needs %zu bytes of LDS per insertion", h.ef_construct, lds1);
        // (a policy may keep static LDS of its own - a decoded row, a rotation buffer -: the dynamic share is what is left of the CU's 160 KiB)
        auto allow = [](const void *kfn) -> int32_t {
            hipFuncAttributes fa;
            QMX_HIP(hipFuncGetAttributes(&fa, kfn));
            QMX_HIP(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)fa.sharedSizeBytes));
            return QMX_OK;
        };
        if (e_sel == 0) {
            static thread_local DeviceOnce attr_once;
            if (attr_once.need()) {
                QMX_TRY(allow(reinterpret_cast<const void *>(k0)));
                attr_once.mark();
            }
        } else if (lds1 > 48 * 1024) {      // a register beam behind a large query entry (TurboQuant over Manhattan: the entry and its hop scratch)
            static thread_local DeviceOnce attr_once28;
            if (attr_once28.need()) {
                QMX_TRY(allow(reinterpret_cast<const void *>(k2)));
                QMX_TRY(allow(reinterpret_cast<const void *>(k8)));
                attr_once28.mark();
            }
        }
        if (grid == 0) {
            int n = 0;
            if (e_sel == 8) QMX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k8, 64, lds1));
            else if (e_sel == 2) QMX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k2, 64, lds1));
            else QMX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k0, 64, lds1));
            *per_cu = n < 1 ? 1 : n;
            return QMX_OK;
        }
        ::qmx::clear_stale_error();
        if (e_sel == 8) hipLaunchKernelGGL(k8, dim3(grid), dim3(64), lds1, st, a, h);
        else if (e_sel == 2) hipLaunchKernelGGL(k2, dim3(grid), dim3(64), lds1, st, a, h);
        else hipLaunchKernelGGL(k0, dim3(grid), dim3(64), lds1, st, a, h);
        QMX_HIP(hipGetLastError());
        return QMX_OK;
    }
    if (grid == 0) {
        int n = 0;
        QMX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, hnsw_build_link_kernel<HI>, 64, 0));
        *per_cu = n < 1 ? 1 : n;
        return QMX_OK;
    }
    ::qmx::clear_stale_error();
    hipLaunchKernelGGL((hnsw_build_link_kernel<HI>), dim3(grid), dim3(64), 0, st, a, h);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

struct HnswBuildLauncher {
    hipStream_t st;
    const HnswBuildArgs *h;
    int phase;
    uint32_t grid;
    int *per_cu;
    template <class P> int32_t row(const ScanArgs &a) const { return launch_hnsw_build_hop<HopRow<P>>(st, a, *h, phase, grid, per_cu); }
    template <class S> int32_t small(const ScanArgs &a) const { return launch_hnsw_build_hop<HopSmall<S>>(st, a, *h, phase, grid, per_cu); }
};

// points = multi-vectors over the segment's inner rows (ScanArgs::mv_offsets), MaxSim between stored points
struct HnswBuildMaxSimLauncher {
    hipStream_t st;
    const HnswBuildArgs *h;
    int phase;
    uint32_t grid;
    int *per_cu;
    template <class P> int32_t row(const ScanArgs &a) const { return launch_hnsw_build_hop<HopMaxSimInternal<HopRow<P>>>(st, a, *h, phase, grid, per_cu); }
    template <class S> int32_t small(const ScanArgs &a) const { return launch_hnsw_build_hop<HopMaxSimInternal<HopSmall<S>>>(st, a, *h, phase, grid, per_cu); }
};

}  // namespace qmx

// hnsw.hpp — device-resident HNSW search: one wavefront walks the graph for one query.
//
// Replaces, for a graph whose links sit in HBM next to the stored block,
//   GraphLayers::search                    lib/segment/src/index/hnsw_index/graph_layers.rs:530-562
//   GraphLayersBase::search_on_level       graph_layers.rs:108-149   (ef-bounded beam, level 0)
//   GraphLayersBase::search_entry_on_level graph_layers.rs:279-317   (greedy descent, levels > 0)
//   GraphLayersBase::search_entry          graph_layers.rs:247-277
//   GraphLayers::get_entry_point           graph_layers.rs:505-528 + entry_points.rs:96-112
//   SearchContext::process_candidate       search_context.rs:32-40
//   FilteredScorer::score_points           point_scorer.rs:265-304   (deleted filter + limit, then score)
//   VisitedListHandle                      index/visited_pool.rs:18-80
// The per-hop `RawScorer::score_points` call of the reference (<= m0 ids per hop) becomes an
// in-kernel gather with the lane policies of the brute-force scan, so a hop never leaves the GPU
// and every score carries the same bits as the scan / the x86 reference.
//
// Mapping
//   * block = 1 wavefront = 1 search at a time; grid = "slots" (CUs x occupancy), each slot loops
//     over queries slot, slot + grid, ...  Many independent searches per CU hide the latency of the
//     dependent loads of a hop (offsets -> links -> visited word -> rows).
//   * the query (tile entry / SQ codes / PQ LUT) is staged in LDS once per search.
//   * `nearest` and `candidates` of SearchContext are ONE sorted register list of 64*E keys
//     (entry e*64+lane) with an "expanded" flag per entry: every candidate the reference can still
//     pop with score >= lower_bound is an element of `nearest` (an evicted candidate is below the
//     bound and ends the loop), so "pop the best candidate" = "best unexpanded entry" and the loop
//     ends when the first ef entries are all expanded.  Identical results whenever scores are
//     distinct; among equal scores the reference's order is BinaryHeap-dependent (unpinned).
//   * visited set = one bit per point in a per-slot HBM bitmap, test-and-set with an L2 atomic;
//     the words a search touched are logged and cleared afterwards (whole-bitmap clear if the
//     log overflows).
//   * links = the plain GraphLinks arrays (graph_links/view.rs: reindex, level_offsets, offsets,
//     neighbors) as serialised by graph_links/serializer.rs:52-176.
#pragma once
#include "custom_combine.hpp"
#include "scan_common.hpp"

namespace qmx {

// bytes of the ACORN walk's list of nodes to explore (one per link of the popped candidate: m0 ids, 64 at least)
__host__ __device__ static inline uint32_t hnsw_acorn_lds(uint32_t acorn, uint32_t m0) { return acorn ? 4u * (((m0 < 64u ? 64u : m0) + 63u) / 64u * 64u) : 0u; }

// ---- hop scorers: LPI lanes score one stored row; the score is valid in the lane with sub == 0 ----
// a policy whose query offset comes from ScanArgs::sq_qoff instead of the query entry's aux block (a stored SQ row as the query)
template <class P, class = void>
struct has_internal_qoff { static constexpr bool value = false; };
template <class P>
struct has_internal_qoff<P, decltype((void)P::INTERNAL_QOFF)> { static constexpr bool value = P::INTERNAL_QOFF; };

// ... or whose query norm comes from ScanArgs::u8_qnorm_* (a stored u8 row as the query of the per-pair cosine)
template <class P, class = void>
struct has_internal_norm { static constexpr bool value = false; };
template <class P>
struct has_internal_norm<P, decltype((void)P::INTERNAL_NORM)> { static constexpr bool value = P::INTERNAL_NORM; };
// a hop scorer whose stored <-> stored score is NOT the score of the search (PQ: LUT of the original vector vs centroid <-> centroid)
template <class H, class = void>
struct is_asymmetric { static constexpr bool value = false; };
template <class H>
struct is_asymmetric<H, decltype((void)H::ASYMMETRIC)> { static constexpr bool value = H::ASYMMETRIC; };

template <class P, class = void>
struct offset_by_caller { static constexpr bool value = false; };
template <class P>
struct offset_by_caller<P, decltype((void)P::OFFSET_BY_CALLER)> { static constexpr bool value = P::OFFSET_BY_CALLER; };

template <class P>
struct HopRow {
    static constexpr int LPI = 8;
    static constexpr bool ADDS_OFFSET = offset_by_caller<P>::value;      // the policy's score lacks the row's offset: hop_score adds hop_scores[j]'s prefilled value
    static constexpr bool INTERNAL_QOFF = has_internal_qoff<P>::value;
    static constexpr bool INTERNAL_NORM = has_internal_norm<P>::value;
    static constexpr bool MULTI = true;     // score_multi<R>: R rows per 8-lane group in one pass
    static __device__ __forceinline__ float score(const ScanArgs &a, const unsigned char *qp, uint32_t id, int sub) {
        return group_score<P>(a, qp, id, sub);
    }
    template <int R>
    static __device__ __forceinline__ void score_multi(const ScanArgs &a, const unsigned char *qp, const uint32_t (&ids)[R], int sub, float (&out)[R]) {
        group_score_multi<P, R>(a, qp, ids, sub, out);
    }
};
template <class S>
struct HopSmall {
    static constexpr int LPI = 1;
    static constexpr bool INTERNAL_QOFF = false;
    static constexpr bool INTERNAL_NORM = false;
    static constexpr bool MULTI = false;
    static __device__ __forceinline__ float score(const ScanArgs &a, const unsigned char *qp, uint32_t id, int) {
        return S::score(qp, reinterpret_cast<const unsigned char *>(a.rows) + (uint64_t)id * a.row_stride, id, a);
    }
};

// Multi-vector points with the MaxSim comparator (MultiMetricQueryScorer, query_scorer/multi_metric_query_scorer.rs + score_max_similarity,
// query_scorer/mod.rs:70-97; quantized: QuantizedMultivectorStorage::score_point_max_similarity, quantized_multivector_storage/mod.rs:339-363):
// the graph's points are multi-vectors, a hop candidate's score is the sum over the query's inner vectors (in order, from 0.0) of the max
// over the point's inner vectors (`if max_sim < sim`, from -inf) of the inner policy's score.  The query entry in LDS is
// [16-byte header: number of inner query vectors, pad, pointer to the first entry][the entries, when they fit the launch's LDS share - else the header points
// at them in global memory (PQ: every inner query vector is a LUT)]; qp points behind the header.
template <class H, class = void>
struct is_maxsim { static constexpr bool value = false; };
template <class H>
struct is_maxsim<H, decltype((void)H::MAXSIM)> { static constexpr bool value = H::MAXSIM; };
template <class HI>
struct HopMaxSim {
    static constexpr int LPI = HI::LPI;
    static constexpr bool INTERNAL_QOFF = false;
    static constexpr bool INTERNAL_NORM = false;
    static constexpr bool MULTI = false;
    static constexpr bool MAXSIM = true;
    static __device__ __forceinline__ float score(const ScanArgs &a, const unsigned char *qp, uint32_t id, int sub) {
        const uint32_t n_tokens = *reinterpret_cast<const uint32_t *>(qp - 16);
        const unsigned char *toks = *reinterpret_cast<const unsigned char *const *>(qp - 8);
        const uint64_t b0 = a.mv_offsets[id], b1 = a.mv_offsets[id + 1];
        float sum = 0.0f;
        for (uint32_t t = 0; t < n_tokens; ++t) {
            const unsigned char *qe = toks + (size_t)t * a.q_stride;
            float max_sim = -__builtin_inff();
            for (uint64_t b = b0; b < b1; ++b) {
                const float sim = HI::score(a, qe, (uint32_t)b, sub);
                if (max_sim < sim) max_sim = sim;
            }
            sum += max_sim;
        }
        return sum;
    }
};

// The insertion searches of a build over multi-vector points whose inner storage cannot turn a stored row into a query (PQ, TurboQuant:
// QuantizedMultivectorStorage::encode_internal_vector -> None, quantized_multivector_storage/mod.rs:458-470; point_scorer.rs:183-218 then takes the query
// scorer of the point's ORIGINAL multi-vector): HopMaxSim over the batch's query entries - the new point's tokens are a.mv_q_tokens entries, a.q_stride
// apart, from `qp` on (hnsw_build.hpp points it at the first; no header).
template <class H, class = void>
struct is_maxsim_q { static constexpr bool value = false; };
template <class H>
struct is_maxsim_q<H, decltype((void)H::MAXSIM_Q)> { static constexpr bool value = H::MAXSIM_Q; };
template <class HI>
struct HopMaxSimQ {
    static constexpr int LPI = HI::LPI;
    static constexpr bool INTERNAL_QOFF = false;
    static constexpr bool INTERNAL_NORM = false;
    static constexpr bool MULTI = false;
    static constexpr bool MAXSIM_Q = true;
    static __device__ __forceinline__ float score(const ScanArgs &a, const unsigned char *qp, uint32_t id, int sub) {
        const uint32_t n_tokens = a.mv_q_tokens;
        const uint64_t b0 = a.mv_offsets[id], b1 = a.mv_offsets[id + 1];
        float sum = 0.0f;
        for (uint32_t t = 0; t < n_tokens; ++t) {
            const unsigned char *qe = qp + (size_t)t * a.q_stride;
            float max_sim = -__builtin_inff();
            for (uint64_t b = b0; b < b1; ++b) {
                const float sim = HI::score(a, qe, (uint32_t)b, sub);
                if (max_sim < sim) max_sim = sim;
            }
            sum += max_sim;
        }
        return sum;
    }
};

// The same points scored stored <-> stored (HNSW build over multi-vector points, hnsw/build.rs:334-341 through FilteredScorer::new_internal:
// MultiMetricQueryScorer::score_internal, multi_metric_query_scorer.rs; quantized: score_internal_max_similarity,
// quantized_multivector_storage/mod.rs:366-393): sum over the inner vectors of point a (in order, from 0.0) of the max over the inner vectors of
// point b (`if sim > max_sim`, from -inf) of the inner storage's score_internal.  The "query entry" of a stored multi-vector is its point id
// (stored_query in hnsw_build.hpp hands it over in the pointer); the inner rows are read from HBM / L2 on every pair.  Not symmetric.
template <class H, class = void>
struct is_maxsim_internal { static constexpr bool value = false; };
template <class H>
struct is_maxsim_internal<H, decltype((void)H::MAXSIM_INTERNAL)> { static constexpr bool value = H::MAXSIM_INTERNAL; };
template <class HI>
struct HopMaxSimInternal {
    static constexpr int LPI = HI::LPI;
    static constexpr bool INTERNAL_QOFF = false;     // the inner rows' offsets / norms are looked up per inner row below
    static constexpr bool INTERNAL_NORM = false;
    static constexpr bool MULTI = false;
    static constexpr bool ASYMMETRIC = true;
    static constexpr bool MAXSIM_INTERNAL = true;
    static __device__ __forceinline__ float score(const ScanArgs &a, const unsigned char *qp, uint32_t id, int sub) {
        const uint32_t p = (uint32_t)reinterpret_cast<uintptr_t>(qp);
        const unsigned char *rows = reinterpret_cast<const unsigned char *>(a.rows);
        const uint64_t a0 = a.mv_offsets[p], a1 = a.mv_offsets[p + 1];
        const uint64_t b0 = a.mv_offsets[id], b1 = a.mv_offsets[id + 1];
        float sum = 0.0f;
        for (uint64_t i = a0; i < a1; ++i) {
            ScanArgs b = a;
            if constexpr (HI::INTERNAL_QOFF) b.sq_qoff = a.row_offsets[i] - a.sq_shift;
            if constexpr (HI::INTERNAL_NORM) {
                b.u8_qnorm_f = a.row_norms_f[i];
                b.u8_qnorm_i = a.row_norms_i[i];
            }
            const unsigned char *qe = rows + i * a.row_stride;
            float max_sim = -__builtin_inff();
            for (uint64_t j = b0; j < b1; ++j) {
                const float sim = HI::score(b, qe, (uint32_t)j, sub);
                if (sim > max_sim) max_sim = sim;
            }
            sum += max_sim;
        }
        return sum;
    }
};

// a hop policy that can drop candidates on an upper bound of their score before the exact scoring (H::prefilter): HopPQ's 8-bit LUT image, pq.hip
template <class H, class = void>
struct hop_adds_offset { static constexpr bool value = false; };
template <class H>
struct hop_adds_offset<H, decltype((void)H::ADDS_OFFSET)> { static constexpr bool value = H::ADDS_OFFSET; };
template <class H, class = void>
struct has_hop_prefilter { static constexpr bool value = false; };
template <class H>
struct has_hop_prefilter<H, decltype((void)H::HOP_PREFILTER)> { static constexpr bool value = H::HOP_PREFILTER; };

// a hop policy that scores a whole hop itself, wave-cooperatively (H::hop): TurboQuant over Manhattan, tq_l1_policy.hpp
template <class H, class = void>
struct is_tql1 { static constexpr bool value = false; };
template <class H>
struct is_tql1<H, decltype((void)H::TQL1)> { static constexpr bool value = H::TQL1; };

// Custom queries as the scorer of the walk (raw_scorer.rs:228-333 builds a CustomQueryScorer / QuantizedCustomQueryScorer / TurboCustomQueryScorer for
// whatever storage the segment has; graph_layers.rs:108-149 walks with whatever scorer it gets): a hop candidate's score is
// query.score_by(|example| inner policy's score(example, candidate)) - the examples of ONE custom query, in flat_iter() order, are what the search stages.
// The query block in LDS is [32-byte CustomHeader][the examples' entries, when they fit: else the header points at them in global memory]; qp points
// behind the header.
struct CustomHeader {
    uint32_t kind, n_a, n_b, coef_first;
    const unsigned char *entries;       // the examples' query entries, a.q_stride apart (LDS or global) ...
    const uint32_t *ex_off;             // ... or, when the examples are multi-vectors of different lengths: byte offset of example e's first token from `entries`
};
static_assert(sizeof(CustomHeader) == 32, "the custom walk's LDS header");
template <class H, class = void>
struct is_custom { static constexpr bool value = false; };
template <class H>
struct is_custom<H, decltype((void)H::CUSTOM)> { static constexpr bool value = H::CUSTOM; };
template <class HI>
struct HopCustom {
    static constexpr int LPI = HI::LPI;
    static constexpr bool INTERNAL_QOFF = false;
    static constexpr bool INTERNAL_NORM = false;
    static constexpr bool MULTI = false;
    static constexpr bool CUSTOM = true;
    static constexpr bool MULTI_EXAMPLES = is_maxsim<HI>::value;      // MultiCustomQueryScorer: every example is a multi-vector scored by MaxSim
    static __device__ __forceinline__ float score(const ScanArgs &a, const unsigned char *qp, uint32_t id, int sub) {
        const CustomHeader *hd = reinterpret_cast<const CustomHeader *>(qp - sizeof(CustomHeader));
        const unsigned char *ent = hd->entries;
        const uint32_t *off = hd->ex_off;
        return custom_score_by(hd->kind, hd->n_a, hd->n_b, a.cq_coefs + hd->coef_first, [&](uint32_t e) {
            return HI::score(a, MULTI_EXAMPLES ? ent + off[e] : ent + (size_t)e * a.q_stride, id, sub);
        });
    }
    // an inner policy that scores a hop as a wave (TurboQuant over Manhattan): the hop against every example in turn - scores parked per example in LDS -
    // then lane j combines candidate j's.  The examples are always staged; behind them [the inner policy's hop scratch][n_examples x 64 floats]
    // (custom_tql1_lds_bytes sizes it; instantiated for such policies only)
    static constexpr bool TQL1 = is_tql1<HI>::value;
    static __device__ __forceinline__ void hop(const ScanArgs &a, const unsigned char *qp, const uint32_t *hop_ids, float *hop_scores, uint32_t k, int lane) {
        const CustomHeader *hd = reinterpret_cast<const CustomHeader *>(qp - sizeof(CustomHeader));
        const uint32_t ne = hd->kind <= QMX_CUSTOM_RECO_SUM_SCORES ? hd->n_a + hd->n_b : hd->n_a + 2 * hd->n_b;
        unsigned char *scratch = const_cast<unsigned char *>(qp) + (size_t)ne * a.q_stride;
        float *ex = reinterpret_cast<float *>(scratch + HI::scratch_bytes(a));
        for (uint32_t e = 0; e < ne; ++e) HI::hop_at(a, hd->entries + (size_t)e * a.q_stride, scratch, hop_ids, ex + (size_t)e * 64, k, lane);
        if ((uint32_t)lane < k)
            hop_scores[lane] = custom_score_by(hd->kind, hd->n_a, hd->n_b, a.cq_coefs + hd->coef_first, [&](uint32_t e) { return ex[(size_t)e * 64 + (uint32_t)lane]; });
        __syncthreads();
    }
};

// bytes of the LDS beam of a walk with max(top, ef) = ef > HNSW_MAX_EF_REG: keys + expanded flags
__host__ __device__ static inline size_t hnsw_beam_lds(uint32_t ef) { return ((size_t)ef * 9 + 15) / 16 * 16; }

// ---- the beam: sorted descending, entry index = e * 64 + lane ------------------------------------
template <int E>
struct Beam {
    uint64_t key[E];
    uint32_t done[E];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int e = 0; e < E; ++e) { key[e] = 0; done[e] = 0; }
    }
    __device__ __forceinline__ uint64_t at(uint32_t idx) const {   // idx uniform
        uint64_t r = 0;
#pragma unroll
        for (int e = 0; e < E; ++e)
            if ((idx >> 6) == (uint32_t)e) r = readlane_u64(key[e], (int)(idx & 63));
        return r;
    }
    // insert nk (uniform, non-zero), keep the first `cap` entries
    __device__ __forceinline__ void insert(uint64_t nk, uint32_t cap, int lane) {
        uint32_t p = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) p += (uint32_t)__popcll(__ballot(key[e] > nk));
        uint64_t carry_k = 0;
        uint32_t carry_d = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            uint64_t up = shfl_up1_u64(key[e]);
            uint32_t upd = (uint32_t)__shfl_up((int)done[e], 1, 64);
            const uint64_t last_k = readlane_u64(key[e], 63);
            const uint32_t last_d = (uint32_t)__builtin_amdgcn_readlane((int)done[e], 63);
            if (lane == 0) { up = carry_k; upd = carry_d; }
            const uint32_t idx = (uint32_t)e * 64 + (uint32_t)lane;
            if (idx == p) { key[e] = nk; done[e] = 0; }
            else if (idx > p) { key[e] = up; done[e] = upd; }
            if (idx >= cap) { key[e] = 0; done[e] = 0; }
            carry_k = last_k;
            carry_d = last_d;
        }
    }
    __device__ __forceinline__ bool done_at(uint32_t idx) const {   // idx uniform
        uint32_t r = 0;
#pragma unroll
        for (int e = 0; e < E; ++e)
            if ((idx >> 6) == (uint32_t)e) r = (uint32_t)__builtin_amdgcn_readlane((int)done[e], (int)(idx & 63));
        return r != 0;
    }
    // best unexpanded entry -> its key (0 if none); marks it expanded
    __device__ __forceinline__ uint64_t pop_best(int lane) {
        uint64_t ck = 0;
        bool found = false;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const uint64_t m = __ballot(key[e] != 0 && done[e] == 0);
            if (!found && m) {
                const int l = __builtin_ctzll(m);
                ck = readlane_u64(key[e], l);
                if (lane == l) done[e] = 1;
                found = true;
            }
        }
        return ck;
    }};

// The same list for ef > 512, kept in LDS (the walk kernel appends `cap` keys + `cap` flag bytes to its dynamic LDS): the same interface, every
// operation wave-cooperative.  insert: the position by a two-level search (64 block ends, then the block), then the tail moves up one entry, 64
// at a time from the end; pop_best scans from a hint below which every entry is expanded.  Slower per operation than the register beam, and meant
// to be: searches this wide are rare.
template <>
struct Beam<0> {
    uint64_t *key;
    unsigned char *done;
    uint32_t len, hint;
    __device__ __forceinline__ void init(unsigned char *lds, uint32_t cap) {
        key = reinterpret_cast<uint64_t *>(lds);
        done = lds + 8 * (size_t)cap;
    }
    __device__ __forceinline__ void clear() { len = 0; hint = 0; }
    __device__ __forceinline__ uint64_t at(uint32_t idx) const { return idx < len ? key[idx] : 0ull; }        // idx uniform
    __device__ __forceinline__ bool done_at(uint32_t idx) const { return idx < len && done[idx] != 0; }
    __device__ __forceinline__ void insert(uint64_t nk, uint32_t cap, int lane) {
        // p = number of keys above nk (descending, distinct keys)
        uint32_t p = 0;
        if (len) {
            const uint32_t stride = (len + 63) / 64;                       // <= 64 while cap <= 4096
            const uint32_t last = (uint32_t)lane * stride + stride - 1;     // the end of the lane's block
            const bool whole = last < len && key[last] > nk;                // the block lies above nk as a whole
            const uint32_t b = (uint32_t)__popcll(__ballot(whole));         // (monotone: the first b blocks)
            const uint32_t i = b * stride + (uint32_t)lane;
            const bool above = (uint32_t)lane < stride && i < len && key[i] > nk;
            p = b * stride + (uint32_t)__popcll(__ballot(above));
        }
        const uint32_t new_len = len < cap ? len + 1 : cap;
        if (p >= new_len) return;                                           // (below a full list: the callers test against at(cap - 1) first)
        for (uint32_t hi = new_len - 1; hi > p;) {                          // entries [p, new_len - 1) move up by one, the last of a full list drops out
            const uint32_t lo = hi - p > 64 ? hi - 63 : p + 1;
            const uint32_t i = lo + (uint32_t)lane;
            uint64_t k = 0;
            unsigned char d = 0;
            if (i <= hi) { k = key[i - 1]; d = done[i - 1]; }
            if (i <= hi) { key[i] = k; done[i] = d; }
            hi = lo - 1;
        }
        if (lane == 0) { key[p] = nk; done[p] = 0; }
        len = new_len;
        if (p < hint) hint = p;
    }
    __device__ __forceinline__ uint64_t pop_best(int lane) {
        for (uint32_t base = hint; base < len; base += 64) {
            const uint32_t i = base + (uint32_t)lane;
            const uint64_t m = __ballot(i < len && done[i] == 0);
            if (m) {
                const uint32_t at = base + (uint32_t)__builtin_ctzll(m);
                if (i == at) done[i] = 1;
                hint = at + 1;
                return key[at];
            }
        }
        hint = len;
        return 0ull;
    }
};

// ---- the visited set of a search, in LDS ----------------------------------------------------------------------------------------------
// One bit per point in a per-slot HBM bitmap costs the walk a third of its time at 10 M points: every test-and-set is an L2 atomic on a random word of a
// 1.25 MB bitmap (5 GB over the slots in flight), i.e. an HBM read-modify-write per link - tools/micro/gather_roof measures the row gathers of a hop at
// 6.2 TB/s alone and at 3.4 TB/s with those atomics beside them, which is where the walk sat (profiles/r5_sq_walk_visited.md).  A search inserts a few
// thousand ids, so its set fits the LDS: 1024 buckets (id & 1023) of eight 16-bit tags ((id >> 10) + 1; 0 = empty, 0xFFFF = taken back) = 16 KiB per search.  test_and_set looks
// the tag up in its bucket (one ds_read_b128), claims the first empty half-word with a compare-and-swap on the word that holds it (lanes of the wave insert
// side by side; a lost race re-reads), and when the bucket is full - about 1 % of the inserts of an ef = 128 search, more for wide ones - falls back to the
// HBM bitmap for that id.  An id is recorded in exactly one of the two places (the bucket is consulted first, and a full bucket never frees a slot while the
// search runs: unset() marks the slot instead of emptying it), so the set is exact: same fresh / visited answers as the bitmap alone, hence the same
// walk.  Graphs of more than 65534 x 1024 points (tags would not fit), ACORN and the reference-heap mode keep the bitmap.
// (One search = one wave = one work-group of 64 threads - `__launch_bounds__(64)` on the walk kernels: the plain 16-byte read of a bucket below races with
// nobody but the lanes of its own wave, whose LDS operations execute in order.  A launch with more than one wave per group would need atomic bucket reads.)
struct LdsVisited {
    uint32_t *tab;      // LDS, 4096 words; nullptr: the bitmap only
    // -> true when `id` was visited before; *in_bitmap: the id lives (now or already) in the HBM bitmap, not in the table
    __device__ __forceinline__ bool test_and_set(uint32_t id, uint32_t *vis, bool *in_bitmap) const {
        const uint32_t bit = 1u << (id & 31);
        *in_bitmap = false;
        if (tab) {
            uint32_t *b = tab + (id & 1023u) * 4u;
            const uint32_t tag = (id >> 10) + 1u;
            for (;;) {
                const uint4 w4 = *reinterpret_cast<const uint4 *>(b);
                const uint32_t w[4] = {w4.x, w4.y, w4.z, w4.w};
                int empty = -1;
                bool found = false;
#pragma unroll
                for (int d = 3; d >= 0; --d) {
                    const uint32_t lo = w[d] & 0xFFFFu, hi = w[d] >> 16;
                    found = found || lo == tag || hi == tag;
                    if (hi == 0) empty = 2 * d + 1;
                    if (lo == 0) empty = 2 * d;
                }
                if (found) return true;
                if (empty < 0) break;                                    // the bucket is full: this id is the bitmap's
                const int d = empty >> 1;
                const uint32_t expected = w[d], desired = expected | (tag << (16 * (empty & 1)));
                if (atomicCAS(&b[d], expected, desired) == expected) return false;
            }
        }
        *in_bitmap = true;
        return (atomicOr(&vis[id >> 5], bit) & bit) != 0;
    }
    // takes back an insert this search made (a link behind the level's limit: the reference never looked at it)
    __device__ __forceinline__ void unset(uint32_t id, uint32_t *vis, bool in_bitmap) const {
        if (in_bitmap || !tab) { atomicAnd(&vis[id >> 5], ~(1u << (id & 31))); return; }
        // The slot is not freed but marked (tag 0xFFFF: no id of a graph this table serves has it): a bucket that was ever full stays full, so an id that
        // went to the bitmap because its bucket was full can never be mistaken for a fresh one by a later lookup that finds room in that bucket.
        uint32_t *b = tab + (id & 1023u) * 4u;
        const uint32_t tag = (id >> 10) + 1u;
        for (int d = 0; d < 4; ++d) {
            for (;;) {
                const uint32_t w = b[d];
                uint32_t nw = w;
                if ((w & 0xFFFFu) == tag) nw = w | 0xFFFFu;
                else if ((w >> 16) == tag) nw = w | 0xFFFF0000u;
                else break;
                if (atomicCAS(&b[d], w, nw) == w) return;
            }
        }
    }
    __device__ __forceinline__ void clear(int lane) const {               // wave-cooperative; the caller barriers
        if (!tab) return;
        uint4 *t = reinterpret_cast<uint4 *>(tab);
        for (uint32_t i = (uint32_t)lane; i < HNSW_VIS_LDS_BYTES / 16; i += 64) t[i] = make_uint4(0, 0, 0, 0);
    }
};

// ---- option hnsw_reference_heap_order: the reference's two binary heaps, worked by ONE lane in std's exact sift order ---------------------
// `nearest` = FixedLengthPriorityQueue<ScoredPointOffset> = BinaryHeap<Reverse<T>> of at most ef entries (lib/common/common/src/
// fixed_length_priority_queue.rs:20-65; push :47-59 replaces the root only on strict root < value, through PeekMut = sift_down(0));
// `candidates` = BinaryHeap<ScoredPointOffset> (search_context.rs:8-40; pop = swap with the last + sift_down_to_bottom(0) + sift_up).
// ScoredPointOffset orders by OrderedFloat(score) alone, so which of two equal scores a sift moves depends on where they sit in the array:
// reproducing the arrays is the only way to reproduce the reference's lists among equal scores (integer scorers: SQ, u8, BQ, 1-bit TQ).
// Every heap operation is a chain of dependent loads of one lane; `nearest` sits in LDS, and so do the first HNSW_REF_CAND_LDS entries of
// `candidates` (round 6: the levels every sift touches; rounds 5's heap lived in HBM whole: a pop was a dozen dependent trips to memory); the rest of
// the array (unbounded in the reference) is a per-slot HBM scratch of ref_cap entries.
struct RefHeaps {
    uint2 *nd;                 // nearest: x = idx, y = score bits
    uint2 *cd;                 // candidates: entries [c_lds, c_cap) live here ...
    uint2 *cl;                 // ... entries [0, c_lds) in LDS
    uint32_t n_len, n_cap, c_len, c_cap, c_lds;
    bool overflow;
    __device__ __forceinline__ uint2 cget(uint32_t i) const { return i < c_lds ? cl[i] : cd[i]; }
    __device__ __forceinline__ void cset(uint32_t i, uint2 v) { if (i < c_lds) cl[i] = v; else cd[i] = v; }
    // OrderedFloat::cmp: NaN is the greatest value and equal to itself
    static __device__ __forceinline__ int of_cmp(float a, float b) {
        if (a < b) return -1;
        if (a > b) return 1;
        if (a == b) return 0;
        const bool an = a != a, bn = b != b;
        if (an && bn) return 0;
        return an ? 1 : -1;
    }
    static __device__ __forceinline__ float sc(uint2 v) { return __uint_as_float(v.y); }
    static __device__ __forceinline__ int rev_cmp(uint2 a, uint2 b) { return of_cmp(sc(b), sc(a)); }      // Reverse<T>
    __device__ __forceinline__ void init(unsigned char *lds, uint32_t ef, uint2 *scratch, uint32_t cap, unsigned char *cand_lds, uint32_t cand_lds_entries) {
        nd = reinterpret_cast<uint2 *>(lds);
        cd = scratch;
        cl = reinterpret_cast<uint2 *>(cand_lds);
        c_lds = cand_lds_entries;
        n_len = 0; n_cap = ef ? ef : 1; c_len = 0; c_cap = cap;
        overflow = false;
    }
    // BinaryHeap::sift_up(0, pos) under Reverse
    __device__ void n_sift_up(uint32_t pos) {
        const uint2 elt = nd[pos];
        while (pos > 0) {
            const uint32_t parent = (pos - 1) / 2;
            if (rev_cmp(elt, nd[parent]) <= 0) break;
            nd[pos] = nd[parent];
            pos = parent;
        }
        nd[pos] = elt;
    }
    // BinaryHeap::sift_down_range(pos, end) under Reverse
    __device__ void n_sift_down_range(uint32_t pos, uint32_t end) {
        const uint2 elt = nd[pos];
        uint32_t child = 2 * pos + 1;
        while (end >= 2 && child <= end - 2) {
            child += rev_cmp(nd[child], nd[child + 1]) <= 0 ? 1u : 0u;
            if (rev_cmp(elt, nd[child]) >= 0) { nd[pos] = elt; return; }
            nd[pos] = nd[child];
            pos = child;
            child = 2 * pos + 1;
        }
        if (end >= 1 && child == end - 1 && rev_cmp(elt, nd[child]) < 0) {
            nd[pos] = nd[child];
            pos = child;
        }
        nd[pos] = elt;
    }
    // FixedLengthPriorityQueue::push -> was the value kept (SearchContext::process_candidate's `was_added`)
    __device__ bool n_push(uint2 v) {
        if (n_len < n_cap) {
            nd[n_len] = v;
            n_sift_up(n_len);
            ++n_len;
            return true;                                   // None
        }
        if (of_cmp(sc(nd[0]), sc(v)) < 0) {
            const uint2 removed = nd[0];
            nd[0] = v;
            n_sift_down_range(0, n_len);
            return removed.x != v.x;                       // Some(removed): removed.idx != score_point.idx
        }
        return false;                                      // Some(value): rejected
    }
    __device__ void c_push(uint2 v) {
        if (c_len == c_cap) { overflow = true; return; }
        uint32_t pos = c_len++;
        while (pos > 0) {                                  // sift_up(0, pos)
            const uint32_t parent = (pos - 1) / 2;
            const uint2 pv = cget(parent);
            if (of_cmp(sc(v), sc(pv)) <= 0) break;
            cset(pos, pv);
            pos = parent;
        }
        cset(pos, v);
    }
    __device__ bool c_pop(uint2 *out) {
        if (c_len == 0) return false;
        uint2 item = cget(--c_len);
        if (c_len > 0) {
            const uint2 root = cget(0);
            // sift_down_to_bottom(0) of `item`, then sift_up from where it landed
            const uint32_t end = c_len;
            uint32_t pos = 0, child = 1;
            while (end >= 2 && child <= end - 2) {
                const uint2 l = cget(child), r = cget(child + 1);
                const bool right = of_cmp(sc(l), sc(r)) <= 0;
                cset(pos, right ? r : l);
                pos = child + (right ? 1u : 0u);
                child = 2 * pos + 1;
            }
            if (child == end - 1) { cset(pos, cget(child)); pos = child; }
            while (pos > 0) {
                const uint32_t parent = (pos - 1) / 2;
                const uint2 pv = cget(parent);
                if (of_cmp(sc(item), sc(pv)) <= 0) break;
                cset(pos, pv);
                pos = parent;
            }
            cset(pos, item);
            item = root;
        }
        *out = item;
        return true;
    }
    // SearchContext::process_candidate (search_context.rs:32-40)
    __device__ void process_candidate(uint32_t idx, float score) {
        const uint2 v = make_uint2(idx, __float_as_uint(score));
        if (n_push(v)) c_push(v);
    }
    // into_iter_sorted(): BinaryHeap::into_sorted_vec under Reverse = descending score, in place
    __device__ uint32_t n_into_sorted() {
        uint32_t end = n_len;
        while (end > 1) {
            --end;
            const uint2 t = nd[0]; nd[0] = nd[end]; nd[end] = t;
            n_sift_down_range(0, end);
        }
        return n_len;
    }
};
constexpr int HNSW_E_REF = -1;       // the template value of the walk's E that selects this mode

__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, off, 64);
        const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), off, 64);
        const uint64_t o = ((uint64_t)hi << 32) | lo;
        v = o > v ? o : v;
    }
    return v;
}

// scores hop_ids[0..k) into hop_scores[0..k); uniform control flow, k <= 64
template <class H, int R>
__device__ __forceinline__ void hop_score_pass(const ScanArgs &a, const unsigned char *qp, const uint32_t *hop_ids, float *hop_scores,
                                               uint32_t base, uint32_t k, int sub, int g) {
    constexpr int IPP = 64 / H::LPI;
    uint32_t ids[R];
    float sc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t j = base + (uint32_t)(r * IPP + g);
        ids[r] = hop_ids[j < k ? j : 0];
    }
    H::template score_multi<R>(a, qp, ids, sub, sc);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t j = base + (uint32_t)(r * IPP + g);
        if (j < k && sub == 0) hop_scores[j] = hop_adds_offset<H>::value ? sc[r] + hop_scores[j] : sc[r];
    }
}

// (policies whose score lacks the row's offset - hop_adds_offset: RowSQX - find it in hop_scores[j]: put there by the caller with the links
// (`have_offsets`), or gathered here from the offsets column)
template <class H>
__device__ __forceinline__ void hop_score(const ScanArgs &a, const unsigned char *qp, const uint32_t *hop_ids,
                                          float *hop_scores, uint32_t k, int lane, bool have_offsets = false) {
    constexpr int IPP = 64 / H::LPI;
    const int sub = lane % H::LPI, g = lane / H::LPI;
    __syncthreads();   // hop_ids written by other lanes
    if constexpr (hop_adds_offset<H>::value) {
        if (!have_offsets) {
            for (uint32_t j = (uint32_t)lane; j < k; j += 64) hop_scores[j] = a.row_offsets[hop_ids[j]];
            __syncthreads();
        }
    }
    if constexpr (is_tql1<H>::value) {      // the policy scores the hop as a wave (it ends on a barrier, like the loop below)
        H::hop(a, qp, hop_ids, hop_scores, k, lane);
        return;
    }
    uint32_t base = 0;
    if constexpr (H::MULTI) {
        // the usual hop (9..32 fresh neighbours) in ONE pass with 2 or 4 rows per lane group: one round of gathers
        for (; base + 2 * IPP < k; base += 4 * IPP) hop_score_pass<H, 4>(a, qp, hop_ids, hop_scores, base, k, sub, g);
        for (; base + IPP < k; base += 2 * IPP) hop_score_pass<H, 2>(a, qp, hop_ids, hop_scores, base, k, sub, g);
    }
    for (; base < k; base += IPP) {
        const uint32_t j = base + (uint32_t)g;
        const bool on = j < k;
        const uint32_t id = hop_ids[on ? j : 0];
        const float s = H::score(a, qp, id, sub);
        if (on && sub == 0) hop_scores[j] = hop_adds_offset<H>::value ? s + hop_scores[j] : s;
    }
    __syncthreads();
}

// One search.  `qp` = the query entry (LDS or global).
template <class H, int E>
__device__ __forceinline__ void hnsw_search_one(const ScanArgs &a, const HnswArgs &h, const unsigned char *qp,
                                                uint32_t *hop_ids, float *hop_scores, uint32_t *vis, uint32_t *vlog,
                                                uint32_t qi, int lane, unsigned char *beam_lds = nullptr, uint32_t *vtab = nullptr, const unsigned char *pq8 = nullptr) {
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint32_t n_scored = 0;

    // ---- get_entry_point: first live entry point, else the live extra point of the highest level ----
    bool have_ep = false;
    uint32_t ep_id = 0, ep_level = 0;
    for (uint32_t base = 0; base < h.n_ep && !have_ep; base += 64) {
        const uint32_t i = base + (uint32_t)lane;
        const bool ok = i < h.n_ep && a.del.live(h.ep_ids[i < h.n_ep ? i : 0]);
        const uint64_t m = __ballot(ok);
        if (m) {
            const uint32_t first = base + (uint32_t)__builtin_ctzll(m);
            ep_id = h.ep_ids[first];
            ep_level = h.ep_levels[first];
            have_ep = true;
        }
    }
    if (!have_ep) {
        for (uint32_t i = 0; i < h.n_xp; ++i) {   // max_by_key(level): the last maximal element wins
            const uint32_t id = h.xp_ids[i], lv = h.xp_levels[i];
            if (a.del.live(id) && (!have_ep || lv >= ep_level)) { ep_id = id; ep_level = lv; have_ep = true; }
        }
    }
    if (!have_ep) {
        if (lane == 0) {
            h.out_counts[qi] = 0;
            if (h.out_scored) h.out_scored[qi] = 0;
            if (h.expanded) h.expanded_cnt[qi] = 0;
        }
        return;
    }
    if (ep_level >= h.n_levels) ep_level = h.n_levels - 1;

    // ---- search_entry: greedy descent over levels ep_level .. 1 ----
    uint32_t cur_id = ep_id;
    float cur_score;
    {
        if (lane == 0) hop_ids[0] = cur_id;
        hop_score<H>(a, qp, hop_ids, hop_scores, 1, lane);
        cur_score = hop_scores[0];
        n_scored += 1;
    }
    for (uint32_t level = ep_level; level > 0; --level) {
        if (level != ep_level) n_scored += 1;   // search_entry_on_level re-scores its entry (same value)
        bool changed = true;
        while (changed) {
            changed = false;
            // a node that has no slot on this level (an inconsistent links file: the host validates what it can see) has no links here
            const uint64_t slot = h.level_offsets[level] + h.reindex[cur_id];
            const bool slot_ok = slot < h.level_offsets[level + 1] && slot + 1 < h.n_offsets;
            uint64_t o0 = slot_ok ? h.offsets[slot] : 0, o1 = slot_ok ? h.offsets[slot + 1] : 0;
            if (o1 > h.n_neighbors) o1 = h.n_neighbors;
            if (o0 > o1) o0 = o1;
            uint32_t remaining = h.m;   // filter_truncate limit = level_m
            for (uint64_t base = o0; base < o1 && remaining > 0; base += 64) {
                const uint64_t i = base + (uint64_t)lane;
                const bool on = i < o1;
                const uint32_t id = on ? h.neighbors[i] : 0;
                bool keep = on && id < h.n_points && a.del.live(id);
                const uint64_t mask = __ballot(keep);
                const uint32_t rank = (uint32_t)__popcll(mask & lt_mask);
                keep = keep && rank < remaining;
                uint32_t k = (uint32_t)__popcll(mask);
                if (k > remaining) k = remaining;
                remaining -= k;
                __syncthreads();
                if (keep) hop_ids[rank] = id;
                hop_score<H>(a, qp, hop_ids, hop_scores, k, lane);
                // sequential `if score > current.score` over the batch == first maximum above current
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
                n_scored += k;
            }
        }
    }

    // ---- search_on_level(level 0, ef) ----
    const uint32_t ef = h.ef > h.top ? h.ef : h.top;
    if constexpr (E == HNSW_E_REF) {
        // ---- option hnsw_reference_heap_order: search_on_level with the reference's own heaps (plain walk only) ----
        __syncthreads();                    // (the previous search of this block is done with the LDS heap)
        RefHeaps rh;
        rh.init(beam_lds, ef, h.ref_cands + (uint64_t)blockIdx.x * h.ref_cap, h.ref_cap, beam_lds + hnsw_beam_lds(ef), HNSW_REF_CAND_LDS);
        uint32_t log_cnt = 0, n_pop = 0;
        const LdsVisited lv{vtab};             // (round 6: the visited set in LDS here too - an exact set either way, hnsw.hpp LdsVisited)
        if (lv.tab) {
            if (lane == 0) { bool in_bm; lv.test_and_set(cur_id, vis, &in_bm); }
        } else {
            if (lane == 0) {
                atomicOr(&vis[cur_id >> 5], 1u << (cur_id & 31));
                vlog[0] = cur_id >> 5;
            }
            log_cnt = 1;
        }
        if (lane == 0) rh.process_candidate(cur_id, cur_score);
        while (true) {
            uint32_t ok = 0, c_idx = 0, c_bits = 0;
            if (lane == 0) {
                uint2 c;
                if (rh.c_pop(&c)) {
                    const float lower_bound = rh.n_len ? RefHeaps::sc(rh.nd[0]) : -3.40282347e+38f;     // ScoreType::min_value()
                    if (!(RefHeaps::sc(c) < lower_bound)) { ok = 1; c_idx = c.x; c_bits = c.y; }
                }
            }
            ok = (uint32_t)__builtin_amdgcn_readfirstlane((int)ok);
            if (!ok) break;
            const uint32_t cand = (uint32_t)__builtin_amdgcn_readfirstlane((int)c_idx);
            if (h.pops) {
                if (lane == 0 && n_pop < h.pop_cap) {
                    qmx_scored_point p;
                    p.idx = cand;
                    p.score = __uint_as_float(c_bits);
                    h.pops[(uint64_t)qi * h.pop_cap + n_pop] = p;
                }
                ++n_pop;
            }
            uint64_t o0 = 0, o1 = 0;
            uint32_t packed_id = 0;
            if (h.l0) {
                const uint32_t *rowp = h.l0 + (uint64_t)cand * h.l0_stride;
                packed_id = (uint32_t)lane + 1 < h.l0_stride ? rowp[lane + 1] : 0;
                o1 = rowp[0];
            } else {
                o0 = h.offsets[cand];
                o1 = h.offsets[(uint64_t)cand + 1];
            }
            uint32_t remaining = h.m0;
            for (uint64_t base = o0; base < o1 && remaining > 0; base += 64) {
                const uint64_t i = base + (uint64_t)lane;
                const bool on = i < o1;
                const uint32_t id = h.l0 ? (on ? packed_id : 0) : (on ? h.neighbors[i] : 0);
                const bool live = on && id < h.n_points && a.del.live(id);
                const uint32_t bit = 1u << (id & 31);
                bool in_bm = true, was_visited = true;
                if (lv.tab) {
                    if (live) was_visited = lv.test_and_set(id, vis, &in_bm);
                } else {
                    const uint32_t old = live ? atomicOr(&vis[id >> 5], bit) : bit;
                    was_visited = (old & bit) != 0;
                }
                bool keep = live && !was_visited;
                const uint64_t mask = __ballot(keep);
                const uint32_t rank = (uint32_t)__popcll(mask & lt_mask);
                uint32_t k = (uint32_t)__popcll(mask);
                if (k > remaining) {
                    if (keep && rank >= remaining) { lv.unset(id, vis, in_bm); keep = false; }
                    k = remaining;
                }
                remaining -= k;
                __syncthreads();
                if (keep) hop_ids[rank] = id;
                {   // the bitmap words this search dirtied (with the LDS table: only the ids of full buckets)
                    const uint64_t bm = __ballot(keep && in_bm);
                    const uint32_t brank = (uint32_t)__popcll(bm & lt_mask);
                    if (keep && in_bm && log_cnt + brank < h.log_cap) vlog[log_cnt + brank] = id >> 5;
                    log_cnt += (uint32_t)__popcll(bm);
                }
                hop_score<H>(a, qp, hop_ids, hop_scores, k, lane);
                {
                    // process_candidate in link order (graph_layers.rs:139-143), by one lane.  A candidate that does not beat the root of a FULL `nearest` at
                    // the start of the hop cannot beat it later in the hop either (the root only rises): push would hand it back (`Some(value)`,
                    // fixed_length_priority_queue.rs:47-59) and nothing changes - those are told apart by all lanes at once, the one lane walks the others
                    const bool full = __builtin_amdgcn_readfirstlane((int)(rh.n_len >= rh.n_cap)) != 0;
                    const uint32_t root_bits = (uint32_t)__builtin_amdgcn_readfirstlane((int)(full ? rh.nd[0].y : 0u));
                    const bool mine = (uint32_t)lane < k;
                    const bool may_enter = mine && (!full || RefHeaps::of_cmp(__uint_as_float(root_bits), hop_scores[mine ? lane : 0]) < 0);
                    uint64_t todo = __ballot(may_enter);
                    if (lane == 0) {
                        while (todo) {
                            const uint32_t j = (uint32_t)__builtin_ctzll(todo);
                            todo &= todo - 1;
                            rh.process_candidate(hop_ids[j], hop_scores[j]);
                        }
                    }
                }
                n_scored += k;
                __syncthreads();
            }
        }
        if (lane == 0) {
            const uint32_t n = rh.n_into_sorted();
            const uint32_t cnt = n < h.top ? n : h.top;
            for (uint32_t i = 0; i < cnt; ++i) {
                qmx_scored_point p;
                p.idx = rh.nd[i].x;
                p.score = RefHeaps::sc(rh.nd[i]);
                h.out[(uint64_t)qi * h.top + i] = p;
            }
            h.out_counts[qi] = cnt;
            if (h.out_scored) h.out_scored[qi] = n_scored;
            if (h.pops) h.pop_cnt[qi] = n_pop;
            if (rh.overflow) *a.err_flag = 2;
        }
        __syncthreads();
        lv.clear(lane);
        if (log_cnt <= h.log_cap) {
            for (uint32_t i = (uint32_t)lane; i < log_cnt; i += 64) vis[vlog[i]] = 0;
        } else {
            for (uint64_t w = (uint64_t)lane; w < h.vis_words; w += 64) vis[w] = 0;
        }
        __threadfence();
        return;
    } else {
    Beam<E> beam;
    if constexpr (E == 0) {
        __syncthreads();                    // (the previous search of this block is done with the LDS list)
        beam.init(beam_lds, ef);
    }
    beam.clear();
    uint32_t log_cnt = 0;
    const LdsVisited lv{h.acorn ? nullptr : vtab};      // (ACORN keeps its two bitmaps)
    {
        if (lv.tab) {
            if (lane == 0) { bool in_bm; lv.test_and_set(cur_id, vis, &in_bm); }       // (an empty table: the entry goes in)
        } else {
            if (lane == 0) {
                atomicOr(&vis[cur_id >> 5], 1u << (cur_id & 31));
                vlog[0] = cur_id >> 5;
            }
            log_cnt = 1;
        }
        beam.insert(make_key(cur_score, cur_id), ef, lane);
    }
    if (h.acorn) {
        // ---- search_on_level_acorn (graph_layers.rs:154-243): links that fail the filters are explored instead of scored ----
        const uint32_t half = (uint32_t)(h.vis_words / 2);           // vis = hop1_visited_list, vis + half = hop2_visited_list
        uint32_t *to_explore = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(hop_ids) + 8 * (size_t)h.hop_cap);
        const uint32_t hop_limit = h.m0;                                // hop1_limit = hop2_limit = get_m(0)
        auto mask_le = [](int l) -> uint64_t { return l >= 63 ? ~0ull : ((2ull << l) - 1ull); };
        auto nth_set = [](uint64_t m, uint32_t nth) -> int {            // lane of the nth (1-based) set bit
            for (uint32_t i = 1; i < nth; ++i) m &= m - 1;
            return __builtin_ctzll(m);
        };
        // the links of `node` on level 0, 64 per call: (id, in range)
        auto link_of = [&](uint32_t node, uint64_t base, uint64_t o1, bool *on) -> uint32_t {
            const uint64_t i = base + (uint64_t)lane;
            *on = i < o1;
            if (h.l0) return *on ? h.l0[(uint64_t)node * h.l0_stride + 1 + i] : 0;
            return *on ? h.neighbors[i] : 0;
        };
        auto log_words = [&](bool mine, uint32_t word) {                 // remember the bitmap words this search dirtied
            const uint64_t m = __ballot(mine);
            const uint32_t r = (uint32_t)__popcll(m & lt_mask);
            if (mine && log_cnt + r < h.log_cap) vlog[log_cnt + r] = word;
            log_cnt += (uint32_t)__popcll(m);
        };
        uint32_t n_score = 0;
        // the points collected so far are scored and offered to the beam, in order: once per popped candidate - or earlier, when the next explored node's
        // links might not fit the hop buffer (m0 > 64: up to m0 (m0 + 1) points per candidate; process_candidate touches the beam only, so scoring a
        // prefix early changes nothing)
        auto flush_scores = [&]() {
            hop_score<H>(a, qp, hop_ids, hop_scores, n_score, lane);
            for (uint32_t base = 0; base < n_score; base += 64) {
                const uint32_t j = base + (uint32_t)lane;
                const uint64_t mykey = j < n_score ? make_key(hop_scores[j], hop_ids[j]) : 0;
                uint64_t mm = __ballot(mykey > beam.at(ef - 1));
                while (mm) {
                    const int src = __builtin_ctzll(mm);
                    mm &= mm - 1;
                    const uint64_t nk = readlane_u64(mykey, src);
                    if (nk > beam.at(ef - 1)) beam.insert(nk, ef, lane);
                }
            }
            n_scored += n_score;
            n_score = 0;
            __syncthreads();
        };
        while (true) {
            const uint64_t ck = beam.pop_best(lane);
            if (ck == 0) break;
            const uint32_t cand = key_idx(ck);
            uint32_t n_explore = 0;
            __syncthreads();
            {   // 1-hop neighbours (:196-211): every unvisited link is marked; passing ones are scored (at most hop_limit), the others explored
                uint64_t o0, o1;
                if (h.l0) { o0 = 0; o1 = h.l0[(uint64_t)cand * h.l0_stride]; } else { o0 = h.offsets[cand]; o1 = h.offsets[(uint64_t)cand + 1]; }
                bool broke = false;
                for (uint64_t base = o0; base < o1 && !broke; base += 64) {
                    bool on;
                    const uint32_t id = link_of(cand, base, o1, &on);
                    const bool valid = on && id < h.n_points;
                    const uint32_t bit = 1u << (id & 31);
                    const uint32_t old = valid ? atomicOr(&vis[id >> 5], bit) : bit;      // check_and_update_visited
                    const bool lv = valid ? a.del.live(id) : false;                        // filters().check_vector(hop1), in flight with it
                    const bool fresh = valid && !(old & bit);
                    const bool ok = fresh && lv;
                    const uint64_t okm = __ballot(ok), frm = __ballot(fresh);
                    int brk = 64;
                    const uint32_t need = hop_limit - n_score;
                    if ((uint32_t)__popcll(okm) >= need) { brk = nth_set(okm, need); broke = true; }
                    const bool processed = lane <= brk;
                    if (fresh && !processed) atomicAnd(&vis[id >> 5], ~bit);              // links behind the break were never looked at
                    log_words(fresh && processed, id >> 5);
                    const uint64_t keep = mask_le(brk);
                    if (ok && processed) hop_ids[n_score + (uint32_t)__popcll(okm & lt_mask)] = id;
                    const uint64_t exm = frm & ~okm & keep;
                    if (fresh && !ok && processed) to_explore[n_explore + (uint32_t)__popcll(exm & lt_mask)] = id;
                    n_score += (uint32_t)__popcll(okm & keep);
                    n_explore += (uint32_t)__popcll(exm);
                }
            }
            __syncthreads();
            // 2-hop neighbours (:213-235): the links of every explored node, at most hop_limit scored per node.  The walk is a chain of
            // dependent memory round trips, so per node they are cut to two: the links of node e + 1 are fetched while node e is
            // resolved, and the three lookups of a link (hop-1 bit, hop-2 test-and-set, filter bits) go out together - a hop-2 bit
            // set for a link that turns out to sit in the hop-1 list is taken back, as the reference never sets it.
            uint32_t nx_id = 0;
            uint64_t nx_o1 = 0;
            bool nx_on = false;
            auto fetch_links = [&](uint32_t node) {
                if (h.l0) { nx_o1 = h.l0[(uint64_t)node * h.l0_stride]; } else { nx_o1 = h.offsets[(uint64_t)node + 1] - h.offsets[node]; }
                const uint64_t o0 = h.l0 ? 0 : h.offsets[node];
                // validity against the real count is applied when the node is resolved; the load itself stays inside the table
                const uint64_t lim = h.l0 ? (uint64_t)(h.l0_stride - 1) : (h.n_neighbors > o0 ? h.n_neighbors - o0 : 0);
                nx_id = link_of(node, o0, o0 + (lim < 64 ? lim : 64), &nx_on);
            };
            if (n_explore) fetch_links(to_explore[0]);
            for (uint32_t e = 0; e < n_explore; ++e) {
                if (n_score + hop_limit > h.hop_cap) flush_scores();
                const uint32_t node = to_explore[e];
                const uint64_t cnt = nx_o1;
                uint32_t id0 = nx_id;
                if (e + 1 < n_explore) fetch_links(to_explore[e + 1]);
                const uint64_t o0 = h.l0 ? 0 : h.offsets[node];
                const uint64_t o1 = o0 + cnt;
                uint32_t added = 0;
                bool broke = false;
                for (uint64_t base = o0; base < o1 && !broke; base += 64) {
                    bool on = base + (uint64_t)lane < o1;
                    uint32_t id = id0;
                    if (base != o0) id = link_of(node, base, o1, &on);     // (more than 64 links: not reached with m0 <= 64)
                    const bool valid = on && id < h.n_points;
                    const uint32_t bit = 1u << (id & 31);
                    const uint32_t r1 = valid ? atomicOr(&vis[id >> 5], 0u) : bit;                        // hop1_visited_list.check(hop2)
                    const uint32_t old2 = valid ? atomicOr(&vis[half + (id >> 5)], bit) : bit;            // hop2_visited_list.check_and_update_visited(hop2)
                    const bool lv = valid ? a.del.live(id) : false;                                       // filters().check_vector(hop2)
                    const bool s1 = (r1 & bit) != 0;
                    const bool set2 = valid && !(old2 & bit);            // this lane set the hop-2 bit
                    if (s1 && set2) atomicAnd(&vis[half + (id >> 5)], ~bit);
                    const bool fresh = valid && !s1 && set2;
                    const bool ok = fresh && lv;
                    const uint64_t okm = __ballot(ok);
                    int brk = 64;
                    const uint32_t need = hop_limit - added;
                    if ((uint32_t)__popcll(okm) >= need) { brk = nth_set(okm, need); broke = true; }
                    const bool processed = lane <= brk;
                    if (fresh && !processed) atomicAnd(&vis[half + (id >> 5)], ~bit);
                    log_words(fresh && processed, half + (id >> 5));
                    if (ok && processed) {
                        atomicOr(&vis[id >> 5], bit);                                                     // hop1_visited_list.check_and_update_visited(hop2)
                        hop_ids[n_score + added + (uint32_t)__popcll(okm & lt_mask)] = id;
                    }
                    log_words(ok && processed, id >> 5);
                    added += (uint32_t)__popcll(okm & mask_le(brk));
                }
                n_score += added;
            }
            // score_points_unfiltered + process_candidate, in order (:237-239)
            flush_scores();
        }
    } else {
    // search_on_level_with_vectors: `candidates` also holds what `nearest` evicted before it was expanded; when every entry of the beam is
    // expanded the reference pops the best of those: one whose score EQUALS the lower bound is expanded like any other (the loop breaks on strict
    // `candidate.score < lower_bound` only, graph_layers.rs:354-365), the first one below it has its base vector scored and ends the loop.
    // Every evicted-unexpanded candidate that can still be popped is kept.  Evictions arrive in strictly increasing key order (the entry that leaves is the
    // beam's worst, and whatever left before was worse still), and the bound never falls - so only the evictions of the LATEST score can ever tie with
    // the bound, and an eviction of a higher score retires all earlier ones: they can no longer tie, and only the best of them (ev_old: the last to leave)
    // can still be popped - as the candidate that ends the loop, once the latest group has been popped whole.  The
    // latest-score group lives in ev[0..3] (best first) with its older members on a per-slot stack in global memory (h.ev_spill: ascending keys, so the top of
    // the stack is the next best) - integer link scores (BQ, 1-bit TurboQuant) evict dozens of equal scores.  Among equal scores the pop order is the key's
    // (lower id first) where the reference's is its heap's.
    uint64_t ev[4] = {0, 0, 0, 0}, ev_old = 0;
    uint32_t n_spill = 0, n_offered = 0, n_exact = 0;
    uint64_t *const spill = h.ev_spill ? h.ev_spill + (uint64_t)blockIdx.x * h.ev_cap : nullptr;
    uint32_t n_exp = 0;
    while (true) {
        uint64_t ck = beam.pop_best(lane);
        if (ck == 0) {
            if (h.expanded && ev[0] && key_score(ev[0]) == key_score(beam.at(ef - 1))) {
                ck = ev[0];
                ev[0] = ev[1]; ev[1] = ev[2]; ev[2] = ev[3]; ev[3] = 0;
                if (n_spill) {      // (wave-uniform; agent-scope accesses: the stack is written and read back by this wave through L2)
                    --n_spill;
                    ev[3] = __hip_atomic_load(&spill[n_spill], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else break;
        }
        const uint32_t cand = key_idx(ck);
        if (h.expanded) {
            if (lane == 0 && n_exp < h.xcap) h.expanded[(uint64_t)qi * h.xcap + n_exp] = cand;
            ++n_exp;
        }
        if (h.pops) {     // qmx_hnsw_search_traced: the pop sequence (n_exp doubles as its counter: `expanded` and `pops` are never both set)
            if (lane == 0 && n_exp < h.pop_cap) {
                qmx_scored_point p;
                p.idx = cand;
                p.score = key_score(ck);
                h.pops[(uint64_t)qi * h.pop_cap + n_exp] = p;
            }
            ++n_exp;
        }
        // links of `cand` on level 0: the packed table needs ONE round trip (count and links are independent loads of the
        // same row), the CSR arrays two (offsets, then neighbors)
        uint64_t o0 = 0, o1 = 0;
        uint32_t packed_id = 0, packed_off = 0;
        if (h.l0) {
            const uint32_t *rowp = h.l0 + (uint64_t)cand * h.l0_stride;
            if (h.l0_aux_off) {
                // an SQ graph's wider table (hnsw_pack_level0_aux_kernel): [m0 link slots, 0xFFFFFFFF behind the last link][the vector_offset of every linked
                // row] - no count word, so that m0 = 32 makes a row of 256 bytes: two whole lines
                const uint32_t link_slots = h.l0_aux_off;
                packed_id = (uint32_t)lane < link_slots ? rowp[lane] : 0xFFFFFFFFu;
                if constexpr (hop_adds_offset<H>::value) packed_off = (uint32_t)lane < link_slots ? rowp[link_slots + (uint32_t)lane] : 0u;
                o1 = (uint64_t)__popcll(__ballot(packed_id != 0xFFFFFFFFu));
            } else {
                packed_id = (uint32_t)lane + 1 < h.l0_stride ? rowp[lane + 1] : 0;
                o1 = rowp[0];
            }
        } else {
            o0 = h.offsets[cand];
            o1 = h.offsets[(uint64_t)cand + 1];
        }
        uint32_t remaining = h.m0;
        for (uint64_t base = o0; base < o1 && remaining > 0; base += 64) {
            const uint64_t i = base + (uint64_t)lane;
            const bool on = i < o1;
            const uint32_t id = h.l0 ? (on ? packed_id : 0) : (on ? h.neighbors[i] : 0);
            const bool live = on && id < h.n_points && a.del.live(id);
            const uint32_t bit = 1u << (id & 31);
            bool in_bm = true, was_visited = true;
            if (lv.tab) {
                if (live) was_visited = lv.test_and_set(id, vis, &in_bm);
            } else {
                const uint32_t old = live ? atomicOr(&vis[id >> 5], bit) : bit;
                was_visited = (old & bit) != 0;
            }
            bool keep = live && !was_visited;
            const uint64_t mask = __ballot(keep);
            const uint32_t rank = (uint32_t)__popcll(mask & lt_mask);
            uint32_t k = (uint32_t)__popcll(mask);
            if (k > remaining) {   // more links than level_m: the reference scores only the first `limit`
                if (keep && rank >= remaining) { lv.unset(id, vis, in_bm); keep = false; }
                k = remaining;
            }
            remaining -= k;
            __syncthreads();
            if (keep) {
                hop_ids[rank] = id;
                if constexpr (hop_adds_offset<H>::value) hop_scores[rank] = __uint_as_float(packed_off);      // (read only when the row carried it: below)
            }
            {   // the bitmap words this search dirtied (with the LDS table: only the ids of full buckets)
                const uint64_t bm = __ballot(keep && in_bm);
                const uint32_t brank = (uint32_t)__popcll(bm & lt_mask);
                if (keep && in_bm && log_cnt + brank < h.log_cap) vlog[log_cnt + brank] = id >> 5;
                log_cnt += (uint32_t)__popcll(bm);
            }
            n_scored += k;
            if constexpr (has_hop_prefilter<H>::value) {      // candidates that cannot beat the beam's worst entry leave here, unscored (the same walk: see H::prefilter)
                if (pq8) {
                    n_offered += k;
                    k = H::prefilter(a, pq8, hop_ids, k, beam.at(ef - 1), lane);
                    n_exact += k;
                }
            }
            hop_score<H>(a, qp, hop_ids, hop_scores, k, lane, h.l0 != nullptr && h.l0_aux_off != 0);
            const uint64_t mykey = (uint32_t)lane < k ? make_key(hop_scores[lane], hop_ids[lane]) : 0;
            uint64_t mm = __ballot(mykey > beam.at(ef - 1));
            while (mm) {
                const int src = __builtin_ctzll(mm);
                mm &= mm - 1;
                const uint64_t nk = readlane_u64(mykey, src);
                const uint64_t last = beam.at(ef - 1);
                if (nk > last) {
                    if (h.expanded && last != 0 && !beam.done_at(ef - 1)) {      // an unexpanded entry leaves `nearest`: it stays in `candidates`
                        if (ev[0] && (last >> 32) != (ev[0] >> 32)) {             // a higher score: the earlier evictions are retired
                            ev_old = ev[0];
                            ev[0] = ev[1] = ev[2] = ev[3] = 0;
                            n_spill = 0;
                        }
                        if (ev[3]) {                                              // the group's oldest member of the four makes room
                            if (spill && n_spill < h.ev_cap) {
                                if (lane == 0) __hip_atomic_store(&spill[n_spill], ev[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                ++n_spill;
                            } else if (lane == 0) *a.err_flag = 2;                // (more equal scores than the stack holds: reported, never silently dropped)
                        }
                        ev[3] = ev[2]; ev[2] = ev[1]; ev[1] = ev[0]; ev[0] = last;
                    }
                    beam.insert(nk, ef, lane);
                }
            }
        }
    }
    if (h.expanded) {
        const uint64_t last_pop = ev[0] ? ev[0] : ev_old;
        if (last_pop) {   // the pop that ends the loop: the best evicted candidate below the bound
            if (lane == 0 && n_exp < h.xcap) h.expanded[(uint64_t)qi * h.xcap + n_exp] = key_idx(last_pop);
            ++n_exp;
        }
        if (lane == 0) h.expanded_cnt[qi] = n_exp;
    }
    if (h.pops && lane == 0) h.pop_cnt[qi] = n_exp;
    if (h.pq_stats && lane == 0 && n_offered) {
        atomicAdd(&h.pq_stats[0], (unsigned long long)n_offered);
        atomicAdd(&h.pq_stats[1], (unsigned long long)n_exact);
    }
    }

    // ---- nearest.into_iter_sorted().take(top) ----
    {
        uint32_t count = 0;
        if constexpr (E == 0) {
            for (uint32_t base = 0; base < h.top; base += 64) {
                const uint32_t idx = base + (uint32_t)lane;
                const uint64_t k = idx < h.top ? beam.at(idx) : 0ull;
                const bool ok = k != 0;
                if (ok) {
                    qmx_scored_point p;
                    p.idx = key_idx(k);
                    p.score = key_score(k);
                    h.out[(uint64_t)qi * h.top + idx] = p;
                }
                count += (uint32_t)__popcll(__ballot(ok));
            }
        } else {
#pragma unroll
        for (int e = 0; e < (E ? E : 1); ++e) {
            const uint32_t idx = (uint32_t)e * 64 + (uint32_t)lane;
            const bool ok = beam.key[e] != 0 && idx < h.top;
            if (ok) {
                qmx_scored_point p;
                p.idx = key_idx(beam.key[e]);
                p.score = key_score(beam.key[e]);
                h.out[(uint64_t)qi * h.top + idx] = p;
            }
            count += (uint32_t)__popcll(__ballot(ok));
        }
        }
        if (lane == 0) {
            h.out_counts[qi] = count;
            if (h.out_scored) h.out_scored[qi] = n_scored;
        }
    }

    // ---- give the visited bitmap (and the LDS table) back all-zero ----
    __syncthreads();
    lv.clear(lane);
    if (log_cnt <= h.log_cap) {
        for (uint32_t i = (uint32_t)lane; i < log_cnt; i += 64) vis[vlog[i]] = 0;
    } else {
        for (uint64_t w = (uint64_t)lane; w < h.vis_words; w += 64) vis[w] = 0;
    }
    __threadfence();
    }   // E != HNSW_E_REF
}

#ifndef QMX_WALK_KERNEL_ATTR
#define QMX_WALK_KERNEL_ATTR
#endif
template <class H, int E, bool QLDS>
__global__ __launch_bounds__(64) QMX_WALK_KERNEL_ATTR void hnsw_search_kernel(const ScanArgs a, const HnswArgs h) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    // [hop ids: hop_cap u32][hop scores: hop_cap f32][ACORN: 64 ids to explore][query entry]
    uint32_t *hop_ids = reinterpret_cast<uint32_t *>(smem);
    float *hop_scores = reinterpret_cast<float *>(smem + 4 * (size_t)h.hop_cap);
    unsigned char *q_lds = smem + 8 * (size_t)h.hop_cap + hnsw_acorn_lds(h.acorn, h.m0);
    unsigned char *beam_lds = q_lds + (QLDS ? ((size_t)h.lds_query_bytes + 15) / 16 * 16 : 0);      // E == 0: the LDS beam behind the query entry (E == HNSW_E_REF: `nearest`)
    uint32_t *vis = h.visited + (uint64_t)blockIdx.x * h.vis_words;
    uint32_t *vlog = h.vis_log + (uint64_t)blockIdx.x * h.log_cap;
    // the search's visited table (LdsVisited), behind everything else: zero once here, every search leaves it zero
    uint32_t *vtab = h.vis_lds ? reinterpret_cast<uint32_t *>(beam_lds + (E <= 0 ? hnsw_beam_lds(h.ef > h.top ? h.ef : h.top) : 0) + (E == HNSW_E_REF ? (size_t)HNSW_REF_CAND_LDS * 8 : 0)) : nullptr;
    if (vtab) {
        LdsVisited{vtab}.clear(lane);
        __syncthreads();
    }
    // Searches are handed out one at a time (h.next_query, set to gridDim.x by the host): a slot that finishes takes the next unstarted search.  With the
    // static stride qi += gridDim.x the launch lasted ceil(nq / slots) searches of the slowest slot - 8 192 searches on 2 304 slots paid for 4 rounds while
    // doing 3.56 (profiles/r5_sq_walk_visited.md).
    auto next_search = [&](uint32_t qi) -> uint32_t {
        if (!h.next_query) return qi + gridDim.x;
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(h.next_query, 1u);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    };
    // HopPQ's 8-bit LUT image of the search (HnswArgs::pq8), behind the visited table
    unsigned char *pq8_lds = nullptr;
    if constexpr (has_hop_prefilter<H>::value) {
        if (h.pq8) pq8_lds = beam_lds + (E <= 0 ? hnsw_beam_lds(h.ef > h.top ? h.ef : h.top) : 0) + h.vis_lds;
    }
    for (uint32_t qi = blockIdx.x; qi < h.nq; qi = next_search(qi)) {
        if constexpr (has_hop_prefilter<H>::value) {
            if (pq8_lds) {
                __syncthreads();
                const uint4 *src = reinterpret_cast<const uint4 *>(h.pq8 + (uint64_t)qi * h.pq8_stride);
                uint4 *dst = reinterpret_cast<uint4 *>(pq8_lds);
                for (uint32_t i = (uint32_t)lane; i < h.pq8_stride / 16; i += 64) dst[i] = src[i];
                __syncthreads();
            }
        }
        if constexpr (is_maxsim<H>::value) {
            // (the header always sits in LDS; the entries follow when the launch's budget holds them, else they are read where they lie)
            const uint32_t t0 = a.mv_qfirst[qi], n_tokens = a.mv_qfirst[qi + 1] - t0;
            const unsigned char *qg = reinterpret_cast<const unsigned char *>(a.queries) + (uint64_t)t0 * a.q_stride;
            const bool fits = 16 + (uint64_t)n_tokens * a.q_stride <= h.lds_query_bytes;
            __syncthreads();
            if (fits) {
                const uint4 *src = reinterpret_cast<const uint4 *>(qg);
                uint4 *dst = reinterpret_cast<uint4 *>(q_lds + 16);
                for (uint32_t i = (uint32_t)lane; i < n_tokens * (a.q_stride / 16); i += 64) dst[i] = src[i];
            }
            if (lane == 0) {
                *reinterpret_cast<uint32_t *>(q_lds) = n_tokens;
                *reinterpret_cast<const unsigned char **>(q_lds + 8) = fits ? q_lds + 16 : qg;
            }
            __syncthreads();
            hnsw_search_one<H, E>(a, h, q_lds + 16, hop_ids, hop_scores, vis, vlog, qi, lane, beam_lds, vtab, pq8_lds);
            continue;
        }
        if constexpr (is_custom<H>::value) {
            // (the header always sits in LDS: launch_hnsw_hop grants at least its 32 bytes; the examples follow when the launch's budget holds them)
            const qmx_custom_query cq = a.cq_desc[qi];
            const uint32_t ne = cq.kind <= QMX_CUSTOM_RECO_SUM_SCORES ? cq.n_a + cq.n_b : cq.n_a + 2 * cq.n_b;
            const unsigned char *qg = reinterpret_cast<const unsigned char *>(a.queries) + (uint64_t)cq.first * a.q_stride;
            if constexpr (H::MULTI_EXAMPLES) {
                // [header][offset table, 16-byte padded][per example: 16-byte MaxSim header (its token count) + its tokens]; the API sized the launch for it
                unsigned char *base = q_lds + sizeof(CustomHeader);
                uint32_t *tab = reinterpret_cast<uint32_t *>(base);
                uint32_t off = (4 * ne + 15u) & ~15u;
                __syncthreads();
                for (uint32_t e = 0; e < ne; ++e) {
                    const uint32_t t0 = a.mv_qfirst[cq.first + e], nt = a.mv_qfirst[cq.first + e + 1] - t0;
                    // (the visited table and the PQ image sit right behind the query entry: an example that did not fit the launch's entry must not be
                    // staged over them - the search is refused instead)
                    if (sizeof(CustomHeader) + (uint64_t)off + 16 + (uint64_t)nt * a.q_stride > h.lds_query_bytes) {
                        if (lane == 0) *a.err_flag = 1;
                        break;
                    }
                    if (lane == 0) {
                        tab[e] = off + 16;
                        *reinterpret_cast<uint32_t *>(base + off) = nt;
                        *reinterpret_cast<const unsigned char **>(base + off + 8) = base + off + 16;
                    }
                    const uint4 *src = reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(a.queries) + (uint64_t)t0 * a.q_stride);
                    uint4 *dst = reinterpret_cast<uint4 *>(base + off + 16);
                    for (uint32_t i = (uint32_t)lane; i < nt * (a.q_stride / 16); i += 64) dst[i] = src[i];
                    off += 16 + nt * a.q_stride;
                }
                if (lane == 0) {
                    CustomHeader *hd = reinterpret_cast<CustomHeader *>(q_lds);
                    hd->kind = cq.kind; hd->n_a = cq.n_a; hd->n_b = cq.n_b; hd->coef_first = cq.coef_first;
                    hd->entries = base;
                    hd->ex_off = tab;
                }
                __syncthreads();
                hnsw_search_one<H, E>(a, h, q_lds + sizeof(CustomHeader), hop_ids, hop_scores, vis, vlog, qi, lane, beam_lds, vtab, pq8_lds);
                continue;
            }
            const bool fits = sizeof(CustomHeader) + (uint64_t)ne * a.q_stride <= h.lds_query_bytes;
            __syncthreads();
            if (fits) {
                const uint4 *src = reinterpret_cast<const uint4 *>(qg);
                uint4 *dst = reinterpret_cast<uint4 *>(q_lds + sizeof(CustomHeader));
                for (uint32_t i = (uint32_t)lane; i < ne * (a.q_stride / 16); i += 64) dst[i] = src[i];
            }
            if (lane == 0) {
                CustomHeader *hd = reinterpret_cast<CustomHeader *>(q_lds);
                hd->kind = cq.kind; hd->n_a = cq.n_a; hd->n_b = cq.n_b; hd->coef_first = cq.coef_first;
                hd->entries = fits ? q_lds + sizeof(CustomHeader) : qg;
                hd->ex_off = nullptr;
            }
            __syncthreads();
            hnsw_search_one<H, E>(a, h, q_lds + sizeof(CustomHeader), hop_ids, hop_scores, vis, vlog, qi, lane, beam_lds, vtab, pq8_lds);
            continue;
        }
        const unsigned char *qg = reinterpret_cast<const unsigned char *>(a.queries) + (uint64_t)qi * a.q_stride;
        if constexpr (QLDS) {
            __syncthreads();
            const uint4 *src = reinterpret_cast<const uint4 *>(qg);
            uint4 *dst = reinterpret_cast<uint4 *>(q_lds);
            const uint32_t q_units = (h.lds_query_bytes < a.q_stride ? h.lds_query_bytes : a.q_stride) / 16;     // (a policy may own scratch behind its entry)
            for (uint32_t i = (uint32_t)lane; i < q_units; i += 64) dst[i] = src[i];
            __syncthreads();
            hnsw_search_one<H, E>(a, h, q_lds, hop_ids, hop_scores, vis, vlog, qi, lane, beam_lds, vtab, pq8_lds);
        } else {
            hnsw_search_one<H, E>(a, h, qg, hop_ids, hop_scores, vis, vlog, qi, lane, beam_lds, vtab, pq8_lds);
        }
    }
}


// occupancy of an instantiation (blocks of one wave per CU) — used by the API to size the scratch
template <class H, int E, bool QLDS>
int32_t hnsw_occupancy_inst(uint32_t lds_query_bytes, size_t hop_lds, int *per_cu) {
    auto kfn = hnsw_search_kernel<H, E, QLDS>;
    static thread_local DeviceOnce attr_once;
    if (attr_once.need()) {
        QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_once.mark();
    }
    const size_t lds = hop_lds + (QLDS ? ((size_t)lds_query_bytes + 15) / 16 * 16 : 0);     // (hop_lds carries the LDS beam of an E == 0 instantiation)
    int n = 0;
    QMX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kfn, 64, lds));
    *per_cu = n < 1 ? 1 : n;
    return QMX_OK;
}

template <class H, int E, bool QLDS>
int32_t launch_hnsw_inst(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid) {
    const uint32_t ef = h.ef > h.top ? h.ef : h.top;
    const size_t lds = 8 * (size_t)h.hop_cap + hnsw_acorn_lds(h.acorn, h.m0) + (QLDS ? ((size_t)h.lds_query_bytes + 15) / 16 * 16 : 0) + (E <= 0 ? hnsw_beam_lds(ef) : 0) +
                       (E == HNSW_E_REF ? (size_t)HNSW_REF_CAND_LDS * 8 : 0) + h.vis_lds + (h.pq8 ? h.pq8_stride : 0);
    QMX_REQUIRE(lds <= 160 * 1024, QMX_ERR_NOT_SUPPORTED, "hnsw walk: %zu bytes of LDS (query entry + a list of %u)", lds, ef);
    ::qmx::clear_stale_error();
    QMX_NOTE_KERNEL((hnsw_search_kernel<H, E, QLDS>));
    hipLaunchKernelGGL((hnsw_search_kernel<H, E, QLDS>), dim3(grid), dim3(64), lds, st, a, h);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

// the policies option hnsw_reference_heap_order is instantiated for: the plain single-vector walks (dense, SQ, PQ, BQ, TurboQuant)
template <class H>
struct ref_heaps_built {
    static constexpr bool value = !is_custom<H>::value && !is_maxsim<H>::value && !is_maxsim_q<H>::value && !is_maxsim_internal<H>::value && !is_tql1<H>::value;
};

// grid == 0: only report the occupancy in *per_cu (no launch)
template <class H>
int32_t launch_hnsw_hop(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu) {
    const uint32_t ef = h.ef > h.top ? h.ef : h.top;
    QMX_REQUIRE(ef >= 1 && ef <= HNSW_MAX_EF, QMX_ERR_NOT_SUPPORTED, "hnsw ef %u not in 1..%u", ef, HNSW_MAX_EF);
    const bool qlds = h.lds_query_bytes > 0;
    if constexpr (is_maxsim<H>::value) QMX_REQUIRE(qlds, QMX_ERR_NOT_SUPPORTED, "the inner vectors of a multi-query must fit the LDS");
    if constexpr (is_custom<H>::value) QMX_REQUIRE(h.lds_query_bytes >= sizeof(CustomHeader), QMX_ERR_OTHER, "the custom walk keeps its header in LDS");
    if (h.ref_heaps) {      // option hnsw_reference_heap_order: the plain walk of the policies a stored graph is walked with
        if constexpr (ref_heaps_built<H>::value) {
            QMX_REQUIRE(!h.acorn && !h.expanded, QMX_ERR_NOT_SUPPORTED, "hnsw_reference_heap_order: the plain walk only (not ACORN, not search_with_vectors)");
            const size_t hop_lds = 8 * (size_t)h.hop_cap + hnsw_beam_lds(ef) + (size_t)HNSW_REF_CAND_LDS * 8 + h.vis_lds + (h.pq8 ? h.pq8_stride : 0);
            if (grid == 0) return qlds ? hnsw_occupancy_inst<H, HNSW_E_REF, true>(h.lds_query_bytes, hop_lds, per_cu) : hnsw_occupancy_inst<H, HNSW_E_REF, false>(0, hop_lds, per_cu);
            return qlds ? launch_hnsw_inst<H, HNSW_E_REF, true>(st, a, h, grid) : launch_hnsw_inst<H, HNSW_E_REF, false>(st, a, h, grid);
        } else {
            set_error("hnsw_reference_heap_order: not built for this scorer (custom queries, multi-vector points, TurboQuant over Manhattan)");
            return QMX_ERR_NOT_SUPPORTED;
        }
    }
    if (ef > HNSW_MAX_EF_REG) {        // the LDS beam: one instantiation per policy, the query entry staged
        QMX_REQUIRE(qlds, QMX_ERR_NOT_SUPPORTED, "hnsw ef %u > %u needs the query entry in LDS (it does not fit)", ef, HNSW_MAX_EF_REG);
        if (grid == 0) return hnsw_occupancy_inst<H, 0, true>(h.lds_query_bytes, 8 * (size_t)h.hop_cap + hnsw_acorn_lds(h.acorn, h.m0) + hnsw_beam_lds(ef) + h.vis_lds + (h.pq8 ? h.pq8_stride : 0), per_cu);
        return launch_hnsw_inst<H, 0, true>(st, a, h, grid);
    }
    if constexpr (is_custom<H>::value || is_maxsim<H>::value) {      // (always staged: no instantiation that reads the entry from global memory)
        if (grid == 0) {
            const size_t hop_lds = 8 * (size_t)h.hop_cap + hnsw_acorn_lds(h.acorn, h.m0) + h.vis_lds + (h.pq8 ? h.pq8_stride : 0);
            return ef <= 128 ? hnsw_occupancy_inst<H, 2, true>(h.lds_query_bytes, hop_lds, per_cu) : hnsw_occupancy_inst<H, 8, true>(h.lds_query_bytes, hop_lds, per_cu);
        }
        return ef <= 128 ? launch_hnsw_inst<H, 2, true>(st, a, h, grid) : launch_hnsw_inst<H, 8, true>(st, a, h, grid);
    } else {
    if (grid == 0) {
        const size_t hop_lds = 8 * (size_t)h.hop_cap + hnsw_acorn_lds(h.acorn, h.m0) + h.vis_lds + (h.pq8 ? h.pq8_stride : 0);
        if (ef <= 128) return qlds ? hnsw_occupancy_inst<H, 2, true>(h.lds_query_bytes, hop_lds, per_cu) : hnsw_occupancy_inst<H, 2, false>(0, hop_lds, per_cu);
        return qlds ? hnsw_occupancy_inst<H, 8, true>(h.lds_query_bytes, hop_lds, per_cu) : hnsw_occupancy_inst<H, 8, false>(0, hop_lds, per_cu);
    }
    if (ef <= 128) return qlds ? launch_hnsw_inst<H, 2, true>(st, a, h, grid) : launch_hnsw_inst<H, 2, false>(st, a, h, grid);
    return qlds ? launch_hnsw_inst<H, 8, true>(st, a, h, grid) : launch_hnsw_inst<H, 8, false>(st, a, h, grid);
    }
}

// launch functor for the per-dtype dispatchers (dispatch_dense / dispatch_sq)
struct HnswLauncher {
    hipStream_t st;
    const HnswArgs *h;
    uint32_t grid;
    int *per_cu;
    template <class P> int32_t row(const ScanArgs &a) const { return launch_hnsw_hop<HopRow<P>>(st, a, *h, grid, per_cu); }
    template <class S> int32_t small(const ScanArgs &a) const { return launch_hnsw_hop<HopSmall<S>>(st, a, *h, grid, per_cu); }
};
struct HnswCustomLauncher {
    hipStream_t st;
    const HnswArgs *h;
    uint32_t grid;
    int *per_cu;
    template <class P> int32_t row(const ScanArgs &a) const { return launch_hnsw_hop<HopCustom<HopRow<P>>>(st, a, *h, grid, per_cu); }
    template <class S> int32_t small(const ScanArgs &a) const { return launch_hnsw_hop<HopCustom<HopSmall<S>>>(st, a, *h, grid, per_cu); }
};
struct HnswCustomMaxSimLauncher {
    hipStream_t st;
    const HnswArgs *h;
    uint32_t grid;
    int *per_cu;
    template <class P> int32_t row(const ScanArgs &a) const { return launch_hnsw_hop<HopCustom<HopMaxSim<HopRow<P>>>>(st, a, *h, grid, per_cu); }
    template <class S> int32_t small(const ScanArgs &a) const { return launch_hnsw_hop<HopCustom<HopMaxSim<HopSmall<S>>>>(st, a, *h, grid, per_cu); }
};
struct HnswMaxSimLauncher {
    hipStream_t st;
    const HnswArgs *h;
    uint32_t grid;
    int *per_cu;
    template <class P> int32_t row(const ScanArgs &a) const { return launch_hnsw_hop<HopMaxSim<HopRow<P>>>(st, a, *h, grid, per_cu); }
    template <class S> int32_t small(const ScanArgs &a) const { return launch_hnsw_hop<HopMaxSim<HopSmall<S>>>(st, a, *h, grid, per_cu); }
};

}  // namespace qmx

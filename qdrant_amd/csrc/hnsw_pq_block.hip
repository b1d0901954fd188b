// hnsw_pq_block.hip — the HNSW walk over PQ codes with ONE WORKGROUP PER SEARCH: the query's LUT in LDS, a controller wave that owns the beam and
// the visited set, and worker waves that fetch and score candidates' links AHEAD of the walk.
//
// Reference: GraphLayers::search (lib/segment/src/index/hnsw_index/graph_layers.rs:530-562) = search_entry (:247-277, greedy descent through
// search_entry_on_level :279-317) + search_on_level (:108-149) with the EncodedVectorsPQ scorer (lib/quantization/src/encoded_vectors_pq.rs:409-443,
// score_point_sse).  The walk itself is hnsw.hpp's (same beam, same order of insertions, same counts of scored points): what changes is WHERE a search
// keeps its working set and how its memory round trips are scheduled.
//
// Why.  A PQ score is m gathers from the query's LUT (m x n_centroids f32: 96 KiB at m = 96).  hnsw_search_kernel<HopPQ> runs one WAVE per search, ~16
// per CU, so the LUTs cannot sit in LDS and every gather is a 4-byte read of a 64-byte sector from beyond L2: 61 x the algorithmic traffic (round-3 PMC)
// and a walk bound by the texture path's gather rate.  With the LUT in LDS (one search per CU) the gathers cost nothing, but a single wave then
// exposes every round trip of a hop (links -> visited word -> code rows): 25 ms per 8 192 searches, measured in round 3.  Here the CU still holds one
// search, and its round trips overlap:
//   * the visited set is a hash set in LDS (no round trip; no bitmap to clean up) - a search that outgrows it restarts on the per-slot HBM bitmap;
//   * worker waves expand the beam's best unexpanded entries speculatively: links row (one round trip), then - for the links not yet visited - the
//     deleted / filter bits and the code rows together (one round trip), then the scores from the LDS LUT in score_point_sse's order.  Scores are pure
//     functions of (query, row): computing them early changes nothing.  The controller COMMITS a candidate when the walk pops it: visited test-and-set,
//     limit, beam insertions, in the reference's order.  The visited set only grows, so a link a worker skipped as visited is visited at commit time,
//     and a link that is fresh at commit time was scored (if a stale view ever says otherwise the controller has the row scored then and there).
// The walk's results are therefore hnsw_search_kernel<HopPQ>'s bit for bit (ids, score bits, scored-point counts): tests/test_gpu_pq.py runs both
// kernels against the oracle's walk.  Traffic: link rows + the code rows of the links that were unvisited when a worker looked (~1.3 x algorithmic).
#include "hnsw.hpp"

namespace qmx {

#ifndef PQB_PROF
#define PQB_PROF 0          // 1 (QMX_TUNING builds only): the controller of block 0 prints where the wall time of its first searches went
#endif
constexpr int PQB_LINKS = 64;                  // links per job (a batch of a node's link list)
constexpr int PQB_MAX_WAVES = 8;
constexpr uint32_t PQB_NONE = 0xFFFFFFFFu;
constexpr uint32_t PQB_SELF = 0xFFFFFFFEu;     // job: score the node itself
constexpr uint32_t PQB_QUIT = 0xFFFFFFFDu;     // job: the search is over

// one worker's mailbox + result block (LDS)
struct PqbSlot {
    uint32_t cmd;              // sequence number of the job posted (controller)
    uint32_t done;             // == cmd when the job's results are complete (worker)
    uint32_t id, level, base;  // the job: links [base, base + 64) of node `id` on `level` (0: packed table + visited pre-filter; > 0: CSR arrays; PQB_SELF / PQB_QUIT)
    uint32_t total;            // links the node has on that level (worker)
    uint32_t n;                // ... of them listed here
    uint32_t pad;
    uint32_t link[PQB_LINKS];
    float score[PQB_LINKS];
    uint32_t flag[PQB_LINKS];  // bit 0: in range, not deleted, allowed; bit 1: scored
    uint32_t cand[PQB_LINKS];  // scratch: the listed links a worker scores, compacted (position in link[])
};

// Everything the waves of a block share lives in LDS and is addressed as LDS (address space 3: ds_read / ds_write / ds_cmpst; through generic pointers the
// polls and the hash probes became flat instructions - the first version of this kernel spent its time there: 45 ms per 8 192 searches against 7.7)
typedef __attribute__((address_space(3))) const float pqb_lds_f32;
typedef __attribute__((address_space(3))) uint32_t pqb_lds_u32;
typedef __attribute__((address_space(3))) PqbSlot pqb_lds_slot;

__device__ __forceinline__ uint32_t pqb_load_acquire(pqb_lds_u32 *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void pqb_store_release(pqb_lds_u32 *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// LDS traffic between the lanes of ONE wave (a lane reads what another lane of its wave wrote): the hardware serves a wave's LDS instructions in order;
// this keeps the compiler from moving them across and waits for the writes
__device__ __forceinline__ void pqb_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// ---- the visited hash set (LDS, open addressing, key = id + 1, never deleted from) ----
__device__ __forceinline__ uint32_t pqb_hash(uint32_t id, uint32_t n) { return (uint32_t)(((uint64_t)(id * 0x9E3779B1u) * n) >> 32); }
__device__ __forceinline__ bool pqb_vh_has(pqb_lds_u32 *vh, uint32_t n, uint32_t id) {
    const uint32_t key = id + 1u;
    uint32_t p = pqb_hash(id, n);
    for (uint32_t probes = 0; probes < n; ++probes) {
        const uint32_t v = __hip_atomic_load(&vh[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (v == key) return true;
        if (v == 0) return false;
        p = p + 1 == n ? 0 : p + 1;
    }
    return false;
}
// true: the id was not in the set (and is now)
__device__ __forceinline__ bool pqb_vh_insert(pqb_lds_u32 *vh, uint32_t n, uint32_t id) {
    const uint32_t key = id + 1u;
    uint32_t p = pqb_hash(id, n);
    for (uint32_t probes = 0; probes < n; ++probes) {
        uint32_t old = 0u;
        if (__hip_atomic_compare_exchange_strong(&vh[p], &old, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return true;
        if (old == key) return false;
        p = p + 1 == n ? 0 : p + 1;
    }
    return false;      // (full: the controller restarts on the bitmap long before)
}

// ---- R code rows per 4-lane group against the LDS LUT: HopPQ::score's arithmetic (pq.hip), lane `sub` owns SSE lane `sub` of score_point_sse ----
template <int R>
__device__ __forceinline__ void pqb_score_rows(const ScanArgs &a, pqb_lds_f32 *lut, const uint32_t (&ids)[R], const bool (&on)[R], int sub, float (&out)[R]) {
    const uint32_t m = a.pq_m, ncent = a.pq_ncent, m4 = m & ~3u;
    const uint8_t *rows = reinterpret_cast<const uint8_t *>(a.rows);
    uint4 w[R][8];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint8_t *codes = rows + (uint64_t)ids[r] * a.row_stride;
#pragma unroll
        for (int k = 0; k < 8; ++k) w[r][k] = (on[r] && (uint32_t)(16 * k) < m4) ? *reinterpret_cast<const uint4 *>(codes + 16 * k) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float l = 0.0f;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            if ((uint32_t)(32 * h) < m4) {
                const uint32_t ws[8] = {w[r][2 * h].x, w[r][2 * h].y, w[r][2 * h].z, w[r][2 * h].w, w[r][2 * h + 1].x, w[r][2 * h + 1].y, w[r][2 * h + 1].z, w[r][2 * h + 1].w};
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t cc = 32 * h + 4 * k;
                    v[k] = cc < m4 ? lut[(cc + (uint32_t)sub) * ncent + ((ws[k] >> (8 * sub)) & 0xFF)] : 0.0f;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if ((uint32_t)(32 * h + 4 * k) < m4) l += v[k];
            }
        }
        const float x = l + dpp_f32<DPP_QUAD_XOR2>(l);          // lane 0: l0 + l2, lane 1: l1 + l3
        float sum = x + dpp_f32<DPP_QUAD_XOR1>(x);              // lane 0: (l0 + l2) + (l1 + l3)
        if (on[r]) {
            const uint8_t *codes = rows + (uint64_t)ids[r] * a.row_stride;
            for (uint32_t c = m4; c < m; ++c) sum += lut[c * ncent + codes[c]];
        }
        out[r] = sum;
    }
}

// ---- a worker wave: serve jobs until the search is over ----
__device__ __forceinline__ void pqb_worker(const ScanArgs &a, const HnswArgs &h, pqb_lds_f32 *lut, pqb_lds_slot *my, pqb_lds_u32 *vh, uint32_t vh_n,
                                           pqb_lds_u32 *gvis_flag, const uint32_t *vis, uint32_t &seen, int lane) {
    while (true) {
        uint32_t c;
        while ((c = pqb_load_acquire(&my->cmd)) == seen) __builtin_amdgcn_s_sleep(1);
        const uint32_t id = my->id, level = my->level, base = my->base;
        if (level == PQB_QUIT) {
            seen = c;
            if (lane == 0) pqb_store_release(&my->done, c);
            return;
        }
        // the links of the batch, one per lane
        uint32_t total = 0, link = 0;
        bool on = false;
        if (level == PQB_SELF) {
            total = 1;
            on = lane == 0;
            link = id;
        } else if (level == 0) {
            const uint32_t *rowp = h.l0 + (uint64_t)id * h.l0_stride;
            const uint32_t i = base + (uint32_t)lane;
            link = i + 1 < h.l0_stride ? rowp[i + 1] : 0;       // (count and links: independent loads of the same row, one round trip)
            total = rowp[0];
            if (total > h.l0_stride - 1) total = h.l0_stride - 1;
            on = i < total;
        } else {
            // a node that has no slot on this level (an inconsistent links file) has no links here
            const uint64_t slot = h.level_offsets[level] + h.reindex[id];
            const bool slot_ok = slot < h.level_offsets[level + 1] && slot + 1 < h.n_offsets;
            uint64_t o0 = slot_ok ? h.offsets[slot] : 0, o1 = slot_ok ? h.offsets[slot + 1] : 0;
            if (o1 > h.n_neighbors) o1 = h.n_neighbors;
            if (o0 > o1) o0 = o1;
            total = (uint32_t)(o1 - o0 > 0xFFFFFFFFull ? 0xFFFFFFFFull : o1 - o0);
            const uint64_t i = o0 + base + (uint64_t)lane;
            on = i < o1;
            link = on ? h.neighbors[i] : 0;
        }
        const bool valid = on && link < h.n_points;
        // level 0: a link the walk has visited already can never be fresh again - no bits, no code row, no score for it
        bool want = valid;
        if (level == 0 && valid) {
            const bool seen_already = *gvis_flag ? ((__hip_atomic_load(&vis[link >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (link & 31)) & 1u) != 0
                                                 : pqb_vh_has(vh, vh_n, link);
            want = !seen_already;
        }
        const bool live = (level == PQB_SELF) ? on : (want ? a.del.live(link) : false);      // (in flight with the code rows below)
        const uint64_t wm = __ballot(want);
        const uint32_t nc = (uint32_t)__popcll(wm);
        if (want) my->cand[__popcll(wm & ((1ull << lane) - 1ull))] = (uint32_t)lane;
        my->link[lane] = link;
        pqb_wave_sync();
        uint32_t fl = 0;
        // scores: 16 rows per pass (4 lanes each), two passes' code rows in flight at once
        const int sub = lane & 3, g = lane >> 2;
        for (uint32_t c0 = 0; c0 < nc; c0 += 32) {
            uint32_t ids[2];
            bool ok[2];
            uint32_t pos[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t ci = c0 + 16u * (uint32_t)r + (uint32_t)g;
                ok[r] = ci < nc;
                pos[r] = ok[r] ? my->cand[ci] : 0u;
                ids[r] = (uint32_t)__shfl((int)link, (int)pos[r], 64);
            }
            float sc[2];
            pqb_score_rows<2>(a, lut, ids, ok, sub, sc);
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (ok[r] && sub == 0) my->score[pos[r]] = sc[r];
        }
        fl = (live ? 1u : 0u) | (want ? 2u : 0u);
        my->flag[lane] = on ? fl : 0u;
        if (lane == 0) {
            my->total = total;
            my->n = total > base ? (total - base < (uint32_t)PQB_LINKS ? total - base : (uint32_t)PQB_LINKS) : 0u;
        }
        seen = c;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) pqb_store_release(&my->done, c);
    }
}

// ---- the controller wave: one search ----
template <int E>
struct PqbController {
    const ScanArgs &a;
    const HnswArgs &h;
    pqb_lds_slot *slots;
    int n_workers;
    pqb_lds_u32 *vh;
    uint32_t vh_n;
    pqb_lds_u32 *gvis_flag;
    uint32_t *vis, *vlog;
    int lane;

    // the workers' state, one worker per lane (lanes >= n_workers: none): read in one go instead of slot by slot
    struct Snap {
        uint32_t id;       // the node the slot holds a level-0 / base-0 result for (or is computing it), PQB_NONE otherwise
        bool idle;
    };
    __device__ __forceinline__ Snap snap() const {
        Snap s{PQB_NONE, false};
        if (lane < n_workers) {
            const uint32_t done = pqb_load_acquire(&slots[lane].done);
            s.idle = done == slots[lane].cmd;
            const bool plain = slots[lane].level == 0 && slots[lane].base == 0;
            s.id = plain ? slots[lane].id : PQB_NONE;
            if (!plain && slots[lane].id != PQB_NONE) s.id = PQB_NONE - 8;     // (holds something else: neither free nor a match)
        }
        return s;
    }
    __device__ __forceinline__ bool idle(int w) const { return pqb_load_acquire(&slots[w].done) == slots[w].cmd; }
    __device__ __forceinline__ void post(int w, uint32_t id, uint32_t level, uint32_t base) {      // worker w must be idle
        if (lane == 0) {
            slots[w].id = id;
            slots[w].level = level;
            slots[w].base = base;
            pqb_store_release(&slots[w].cmd, slots[w].cmd + 1u);
        }
        pqb_wave_sync();
    }
    __device__ __forceinline__ void wait(int w) const {
        while (!idle(w)) __builtin_amdgcn_s_sleep(1);
    }
    // a worker that can take a job now (never `exclude`: the slot whose results the caller is reading): a free one, else an idle one whose speculative
    // result is dropped (the walk asks again if it ever pops that node)
    __device__ __forceinline__ int find_worker(int exclude) const {
        while (true) {
            const Snap s = snap();
            const bool ok = lane < n_workers && lane != exclude && s.idle;
            const uint64_t free_m = __ballot(ok && s.id == PQB_NONE), idle_m = __ballot(ok);
            if (free_m) return __builtin_ctzll(free_m);
            if (idle_m) return __builtin_ctzll(idle_m);
            __builtin_amdgcn_s_sleep(1);
        }
    }
    // synchronous job (descent, second batches, repairs); the result is read by the caller before its next post
    __device__ __forceinline__ pqb_lds_slot *job_sync(uint32_t id, uint32_t level, uint32_t base, int exclude = -1) {
        const int w = find_worker(exclude);
        post(w, id, level == 0 && base == 0 ? level : level, base);
        wait(w);
        if (lane == 0) slots[w].id = PQB_NONE;
        pqb_wave_sync();
        return &slots[w];
    }
    __device__ __forceinline__ bool visited_test_and_set(uint32_t id, bool gvis) {     // per lane; true: fresh
        if (gvis) {
            const uint32_t bit = 1u << (id & 31);
            return (atomicOr(&vis[id >> 5], bit) & bit) == 0;
        }
        return pqb_vh_insert(vh, vh_n, id);
    }

    // returns false when the LDS visited set ran full (the caller restarts the search on the bitmap)
    __device__ __forceinline__ bool run(uint32_t qi, bool gvis, uint32_t vh_limit) {
        const uint64_t lt_mask = (1ull << lane) - 1ull;
        uint32_t n_scored = 0;
#if PQB_PROF
        uint64_t t_start = wall_clock64(), t_wait = 0, t_commit = 0, t_spec = 0, t_descent = 0;
        uint32_t n_hops = 0, n_hit_ready = 0, n_hit_busy = 0, n_miss = 0;
#endif
        // ---- get_entry_point: first live entry point, else the live extra point of the highest level (hnsw.hpp) ----
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
            }
            return true;
        }
        if (ep_level >= h.n_levels) ep_level = h.n_levels - 1;

        // ---- search_entry: greedy descent over levels ep_level .. 1 ----
        uint32_t cur_id = ep_id;
        float cur_score;
        {
            pqb_lds_slot *s = job_sync(cur_id, PQB_SELF, 0);
            cur_score = s->score[0];
            n_scored += 1;
        }
        for (uint32_t level = ep_level; level > 0; --level) {
            if (level != ep_level) n_scored += 1;   // search_entry_on_level re-scores its entry (same value)
            bool changed = true;
            while (changed) {
                changed = false;
                uint32_t remaining = h.m;   // filter_truncate limit = level_m
                uint32_t total = 1;
                for (uint32_t base = 0; base < total && remaining > 0; base += 64) {
                    pqb_lds_slot *s = job_sync(cur_id, level, base);
                    total = s->total;
                    const uint32_t n = s->n;
                    const bool on = (uint32_t)lane < n;
                    bool keep = on && (s->flag[lane] & 1u) != 0;
                    const uint64_t mask = __ballot(keep);
                    const uint32_t rank = (uint32_t)__popcll(mask & lt_mask);
                    keep = keep && rank < remaining;
                    uint32_t k = (uint32_t)__popcll(mask);
                    if (k > remaining) k = remaining;
                    remaining -= k;
                    // sequential `if score > current.score` over the batch == first maximum above current
                    const float sc = keep ? s->score[lane] : 0.0f;
                    uint64_t mk = 0;
                    if (keep && sc > cur_score) mk = ((uint64_t)score_to_ord(sc) << 32) | (uint32_t)(~rank);
                    const uint64_t best = wave_max_u64(mk);
                    if (best) {
                        const uint32_t br = ~(uint32_t)best;
                        const uint64_t who = __ballot(keep && rank == br);
                        const int bl = __builtin_ctzll(who);
                        cur_id = (uint32_t)__builtin_amdgcn_readlane((int)s->link[lane < 64 ? lane : 0], bl);
                        cur_score = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sc), bl));
                        changed = true;
                    }
                    n_scored += k;
                }
            }
        }

#if PQB_PROF
        t_descent = wall_clock64() - t_start;
#endif
        // ---- search_on_level(level 0, ef) ----
        const uint32_t ef = h.ef > h.top ? h.ef : h.top;
        Beam<E> beam;
        beam.clear();
        uint32_t log_cnt = 0, n_visited = 0;
        {
            if (gvis) {
                if (lane == 0) {
                    atomicOr(&vis[cur_id >> 5], 1u << (cur_id & 31));
                    vlog[0] = cur_id >> 5;
                }
                log_cnt = 1;
            } else {
                if (lane == 0) pqb_vh_insert(vh, vh_n, cur_id);
                n_visited = 1;
            }
            beam.insert(make_key(cur_score, cur_id), ef, lane);
        }
        bool ok_run = true;
        while (true) {
            const uint64_t ck = beam.pop_best(lane);
            if (ck == 0) break;
            const uint32_t cand = key_idx(ck);
            // the candidate's result block: posted ahead by an earlier hop, or now
#if PQB_PROF
            const uint64_t t0 = wall_clock64();
            ++n_hops;
#endif
            Snap sn = snap();
            int sw;
            {
                const uint64_t hit = __ballot(sn.id == cand);
#if PQB_PROF
                if (hit) { if (__ballot(sn.id == cand && sn.idle)) ++n_hit_ready; else ++n_hit_busy; } else ++n_miss;
#endif
                if (hit) sw = __builtin_ctzll(hit);
                else {
                    sw = find_worker(-1);
                    post(sw, cand, 0, 0);
                    sn = snap();
                }
            }
            // speculate: the best unexpanded entries of the beam get a worker each, while this hop's loads are in flight
            {
                uint32_t tg[PQB_MAX_WAVES];
                int nt = 0;
                const int want_t = n_workers - 1;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    uint64_t um = __ballot(beam.key[e] != 0 && beam.done[e] == 0);
                    while (um && nt < want_t) {
                        const int l = __builtin_ctzll(um);
                        um &= um - 1;
                        const uint32_t t = key_idx(readlane_u64(beam.key[e], l));
#pragma unroll
                        for (int j = 0; j < PQB_MAX_WAVES; ++j)
                            if (j == nt) tg[j] = t;
                        ++nt;
                    }
                }
                // per worker (lane): is what it holds still wanted?  Results for nodes that are not among the walk's next pops give their workers back
                bool wanted = lane == sw;
#pragma unroll
                for (int j = 0; j < PQB_MAX_WAVES; ++j) wanted = wanted || (j < nt && tg[j] == sn.id);
                const bool mine = lane < n_workers && sn.idle;
                if (mine && !wanted && sn.id != PQB_NONE) {
                    slots[lane].id = PQB_NONE;
                    sn.id = PQB_NONE;
                }
                pqb_wave_sync();
                uint64_t free_m = __ballot(mine && lane != sw && sn.id == PQB_NONE);
#pragma unroll
                for (int j = 0; j < PQB_MAX_WAVES; ++j) {
                    if (j >= nt || !free_m) break;
                    if (__ballot(sn.id == tg[j])) continue;            // already held (or being computed)
                    const int fw = __builtin_ctzll(free_m);
                    free_m &= free_m - 1;
                    post(fw, tg[j], 0, 0);
                    if (lane == fw) sn.id = tg[j];
                }
            }
#if PQB_PROF
            const uint64_t t1 = wall_clock64();
            t_spec += t1 - t0;
#endif
            wait(sw);
#if PQB_PROF
            const uint64_t t2 = wall_clock64();
            t_wait += t2 - t1;
#endif
            // ---- commit: visited test-and-set, limit, scores, beam insertions - in the order of the link list ----
            uint32_t remaining = h.m0;
            pqb_lds_slot *s = &slots[sw];
            uint32_t total = s->total;
            for (uint32_t base = 0; base < total && remaining > 0; base += 64) {
                if (base) {      // (more than 64 links: the next batch, fetched now)
                    if (lane == 0) slots[sw].id = PQB_NONE;
                    __builtin_amdgcn_wave_barrier();
                    s = job_sync(cand, 0, base, sw);
                }
                const uint32_t n = s->n;
                const bool on = (uint32_t)lane < n;
                const uint32_t id = on ? s->link[lane] : 0;
                const uint32_t fl = on ? s->flag[lane] : 0;
                bool keep = on && (fl & 1u) != 0 && visited_test_and_set(id, gvis);
                const uint64_t mask = __ballot(keep);
                const uint32_t rank = (uint32_t)__popcll(mask & lt_mask);
                uint32_t k = (uint32_t)__popcll(mask);
                // (k <= remaining: the packed table holds at most m0 links per row - pq_block_walk_ok)
                remaining -= k < remaining ? k : remaining;
                if (gvis) {
                    if (keep && log_cnt + rank < h.log_cap) vlog[log_cnt + rank] = id >> 5;
                    log_cnt += k;
                } else {
                    n_visited += k;
                }
                float sc = s->score[lane < 64 ? lane : 0];
                // a fresh link without a score (a worker's view of the visited set was ahead of the walk: cannot happen while the set only grows - kept as
                // a repair, not as an assumption): scored now
                uint64_t missing = __ballot(keep && (fl & 2u) == 0);
                while (missing) {
                    const int l = __builtin_ctzll(missing);
                    missing &= missing - 1;
                    const uint32_t mid = (uint32_t)__builtin_amdgcn_readlane((int)id, l);
                    pqb_lds_slot *r = job_sync(mid, PQB_SELF, 0, sw);
                    const float v = r->score[0];
                    if (lane == l) sc = v;
                }
                const uint64_t mykey = keep ? make_key(sc, id) : 0;
                uint64_t mm = __ballot(mykey > beam.at(ef - 1));
                while (mm) {
                    const int src = __builtin_ctzll(mm);
                    mm &= mm - 1;
                    const uint64_t nk = readlane_u64(mykey, src);
                    if (nk > beam.at(ef - 1)) beam.insert(nk, ef, lane);
                }
                n_scored += k;
            }
            if (lane == 0) slots[sw].id = PQB_NONE;
            __builtin_amdgcn_wave_barrier();
#if PQB_PROF
            t_commit += wall_clock64() - t2;
#endif
            if (!gvis && n_visited > vh_limit) { ok_run = false; break; }
        }
        // results of speculation nobody popped: dropped with the search (a running worker is waited for before its slot is reused)
        for (int w = 0; w < n_workers; ++w) {
            wait(w);
            if (lane == 0) slots[w].id = PQB_NONE;
        }
        __builtin_amdgcn_wave_barrier();
        if (!ok_run) return false;

        // ---- nearest.into_iter_sorted().take(top) ----
        uint32_t count = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
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
        if (lane == 0) {
            h.out_counts[qi] = count;
            if (h.out_scored) h.out_scored[qi] = n_scored;
        }
#if PQB_PROF
        if (lane == 0 && blockIdx.x == 0 && qi < 4 * gridDim.x)
            printf("[pqb] q %u: total %.1f us  descent %.1f  hops %u (ready %u busy %u miss %u)  lookup+speculate %.1f  wait %.1f  commit %.1f  scored %u\n", qi,
                   (wall_clock64() - t_start) * 0.01, t_descent * 0.01, n_hops, n_hit_ready, n_hit_busy, n_miss, t_spec * 0.01, t_wait * 0.01, t_commit * 0.01, n_scored);
#endif
        // ---- give the visited bitmap back all-zero (bitmap mode only) ----
        if (gvis) {
            if (log_cnt <= h.log_cap) {
                for (uint32_t i = (uint32_t)lane; i < log_cnt; i += 64) vis[vlog[i]] = 0;
            } else {
                for (uint64_t w = (uint64_t)lane; w < h.vis_words; w += 64) vis[w] = 0;
            }
            __threadfence();
        }
        return true;
    }
};

template <int E>
__global__ __launch_bounds__(PQB_MAX_WAVES * 64) void hnsw_pq_block_kernel(const ScanArgs a, const HnswArgs h, uint32_t vh_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_workers = (int)(blockDim.x >> 6) - 1;
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    lds_byte *lds = (lds_byte *)smem;
    float *lut = reinterpret_cast<float *>(smem);
    pqb_lds_slot *slots = (pqb_lds_slot *)(lds + a.q_stride);
    pqb_lds_u32 *vh = (pqb_lds_u32 *)(lds + a.q_stride + (size_t)n_workers * sizeof(PqbSlot));
    pqb_lds_u32 *gvis_flag = vh + vh_n;
    uint32_t *vis = h.visited + (uint64_t)blockIdx.x * h.vis_words;
    uint32_t *vlog = h.vis_log + (uint64_t)blockIdx.x * h.log_cap;
    const uint32_t vh_limit = (uint32_t)((uint64_t)vh_n * 5 / 8);
    uint32_t seen = 0;
    if (threadIdx.x < (uint32_t)n_workers) {
        slots[threadIdx.x].cmd = 0;
        slots[threadIdx.x].done = 0;
        slots[threadIdx.x].id = PQB_NONE;
        slots[threadIdx.x].level = 0;
        slots[threadIdx.x].base = 0;
    }
    for (uint32_t qi = blockIdx.x; qi < h.nq; qi += gridDim.x) {
        __syncthreads();                                   // (the previous search is over: its workers have left their loops)
        {   // the query's LUT -> LDS; an empty visited set
            const uint4 *src = reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(a.queries) + (uint64_t)qi * a.q_stride);
            uint4 *dst = reinterpret_cast<uint4 *>(lut);
            for (uint32_t i = threadIdx.x; i < a.q_stride / 16; i += blockDim.x) dst[i] = src[i];
            for (uint32_t i = threadIdx.x; i < vh_n; i += blockDim.x) vh[i] = 0;
            if (threadIdx.x == 0) *gvis_flag = 0;
        }
        __syncthreads();
        if (wave == 0) {
            PqbController<E> c{a, h, slots, n_workers, vh, vh_n, gvis_flag, vis, vlog, lane};
            if (!c.run(qi, false, vh_limit)) {             // the LDS set ran full: once more, on the per-slot bitmap in HBM
                if (lane == 0) pqb_store_release(gvis_flag, 1u);
                __builtin_amdgcn_wave_barrier();
                c.run(qi, true, 0);
            }
            for (int w = 0; w < n_workers; ++w) {
                c.wait(w);
                c.post(w, 0, PQB_QUIT, 0);
            }
        } else {
            pqb_worker(a, h, (pqb_lds_f32 *)(lds), &slots[wave - 1], vh, vh_n, gvis_flag, vis, seen, lane);
        }
    }
}

// LDS of a launch: the LUT, the workers' slots, the visited set (as large as fits: a set of 8 192 entries serves ef = 128 on 10 M points) + a flag word
static size_t pqb_lds_bytes(const ScanArgs &a, int waves, uint32_t *vh_n_out) {
    const size_t fixed = (size_t)a.q_stride + (size_t)(waves - 1) * sizeof(PqbSlot) + 16;
    const size_t room = fixed < 160 * 1024 ? 160 * 1024 - fixed : 0;
    uint32_t vh_n = (uint32_t)std::min<size_t>(room / 4, a.q_stride > 64 * 1024 ? 14336 : 8192) / 64 * 64;
    const int64_t cap = option(OPT_HNSW_PQ_BLOCK_SET);
    if (cap >= 64 && (uint64_t)cap < vh_n) vh_n = (uint32_t)cap / 64 * 64;
    if (vh_n_out) *vh_n_out = vh_n;
    return fixed + (size_t)vh_n * 4;
}

bool pq_block_walk_ok(const ScanArgs &a, const HnswArgs &h) {
    const uint32_t ef = h.ef > h.top ? h.ef : h.top;
    uint32_t vh_n = 0;
    (void)pqb_lds_bytes(a, PQB_MAX_WAVES, &vh_n);
    return !h.acorn && !h.expanded && !a.cq_desc && !a.mv_offsets && h.l0 != nullptr && h.l0_stride >= 2 && h.l0_stride - 1 <= h.m0 && ef >= 1 &&
           ef <= HNSW_MAX_EF_REG && a.pq_m >= 4 && a.pq_m <= 128 && a.row_stride % 16 == 0 && (reinterpret_cast<uintptr_t>(a.rows) & 15) == 0 && a.q_stride % 16 == 0 &&
           (size_t)a.pq_m * a.pq_ncent * 4 <= a.q_stride && vh_n >= 64 && h.n_points < 0xFFFFFFF0u;
}

template <int E>
static int32_t launch_pqb_inst(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu, int waves) {
    auto kfn = hnsw_pq_block_kernel<E>;
    static thread_local DeviceOnce attr_once;
    if (attr_once.need()) {
        QMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_once.mark();
    }
    uint32_t vh_n = 0;
    const size_t lds = pqb_lds_bytes(a, waves, &vh_n);
    if (grid == 0) {
        int n = 0;
        QMX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kfn, waves * 64, lds));
        *per_cu = n < 1 ? 1 : n;
        return QMX_OK;
    }
    ::qmx::clear_stale_error();
    QMX_NOTE_KERNEL(kfn);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(waves * 64), lds, st, a, h, vh_n);
    QMX_HIP(hipGetLastError());
    return QMX_OK;
}

int32_t launch_hnsw_pq_block(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu, int waves) {
    QMX_REQUIRE(pq_block_walk_ok(a, h), QMX_ERR_NOT_SUPPORTED, "the block-per-search PQ walk does not serve this launch");
    if (waves < 3) waves = 3;
    if (waves > PQB_MAX_WAVES) waves = PQB_MAX_WAVES;
    const uint32_t ef = h.ef > h.top ? h.ef : h.top;
    return ef <= 128 ? launch_pqb_inst<2>(st, a, h, grid, per_cu, waves) : launch_pqb_inst<8>(st, a, h, grid, per_cu, waves);
}

}  // namespace qmx

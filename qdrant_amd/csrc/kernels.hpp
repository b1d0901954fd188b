// kernels.hpp — host-callable launchers of the HIP kernels (internal to libqdrant_amd.so).
#pragma once
#include "common.hpp"

namespace qmx {

constexpr int MAX_QT = 16;         // queries scored per pass of the stored block
constexpr int SCAN_BLOCK = 512;    // 8 wavefronts share one LDS copy of the query tile
constexpr int MAX_TOP_FAST = 64;   // register-resident per-wave list: one entry per lane

// What a scan / gather launch needs.  Plain data, passed by value to the kernels.
struct TqEc {                  // TQ+ symmetric scoring (score_symmetric_ec): i16 weights D'^2 per coordinate, their scale, <M, M>, the rows' xm column
    const int16_t *weights;
    const float *xm;
    float weight_scale, mm_const;
};
struct ScanArgs {
    const void *rows;          // stored block (reference row layout for the dtype)
    uint64_t n_rows;
    uint64_t row_stride;       // bytes
    uint32_t dim;              // elements per row consumed by the metric (actual_dim for SQ)
    uint32_t nseg;             // full 128-byte segments of the SIMD body
    uint32_t rem_pieces;       // 16-byte pieces of the SIMD body in the last, partial segment (0 = none)
    uint32_t tail_start;       // first element of the reference's scalar tail (== dim when none)
    uint32_t aux_off;          // byte offset of the per-query aux block inside a query tile entry
    uint32_t nq;               // live queries in this tile (<= QT)
    const void *queries;       // device, [QT][q_stride] bytes, preprocessed + cast (+ aux)
    uint32_t q_stride;         // bytes between queries
    uint32_t top;
    const uint32_t *ids;       // candidate list or nullptr (= rows 0..n_cand)
    uint64_t n_cand;           // candidates to visit
    DeletedView del;
    uint64_t *partial;         // [grid][partial_qt][top] keys (top-k mode)
    uint32_t partial_qt;       // query stride of `partial` (the small-row kernel; the tiled one uses its QT)
    const uint64_t *gthr;      // [QT] keys or nullptr (scan_mfma16.hip only): a key whose score is a lower bound of the query's final k-th best
                               // score (0 = none): the k-th best of a pre-scan of a prefix of the block (api_search.hip search_enqueue)
    const uint64_t *key_bound; // [QT] exclusive upper bound on accepted keys (top > 64 runs in passes of 64), or nullptr
    float *scores;             // [nq][n_cand] (score mode)
    uint64_t scores_stride;    // elements between queries in `scores`
    int *err_flag;             // set to 1 on an out-of-range id
    const int *run_if;         // nullptr, or: the kernel returns at once unless *run_if != 0 (the exact scan behind the split prefilter, scan_split.hip)
    const uint32_t *q_map;     // pq_scan_kernel only: list q of the launch scores query q_map[q] of the batch (0xFFFFFFFF: a padding slot), or nullptr
    // SQ
    float sq_multiplier;
    const float *row_offsets;  // SQ: per-row f32 offset (SoA copy) or nullptr when inline in rows
    float sq_shift;            // SQ: MetadataInt8::get_shift (encoded_vectors_u8.rs:116-134), used when a stored row is the query
    float sq_qoff;             // SQ, stored row as the query (HNSW build): its query offset = vector_offset - shift (:105-114, 715-728)
    uint32_t flags;            // QMX_SEG_U8_SCALAR_ORDER ...
    // u8 cosine with a STORED ROW as the query (HNSW build): the row's own norm stands in for the aux block of a query entry
    const float *row_norms_f;  // [n] sum of squares of every stored u8 row, in the AVX2 leaf's order (or nullptr)
    const int32_t *row_norms_i; // [n] the same as one exact i32 (scalar-order leaf)
    float u8_qnorm_f;          // set per query row by hnsw_build.hpp query_args
    int32_t u8_qnorm_i;
    // PQ
    uint32_t pq_m, pq_ncent;
    const float *pq_centroids; // [ncent][pq_dim] the codebook (the LUT-free walk, pq.hip HopPQDirect), or nullptr
    uint32_t pq_dim, pq_chunk, pq_kind;      // ... its geometry: floats per vector, per chunk; 0 dot / cosine, 1 Manhattan, 2 Euclid (PqGeom::kind)
    const float *pq_pair;      // [m][ncent][ncent] chunk distances between centroids = the terms of EncodedVectorsPQ::score_internal, or nullptr
    uint32_t pq_invert;
    // BQ: calculate_metric (encoded_vectors_binary.rs:766-810)
    uint32_t bq_dim;           // original dimension
    uint32_t bq_flip;          // 0: zeros - xor (every distance with its own invert), 1: xor - zeros
    uint32_t bq_qbits;         // bit planes per query value: 1 (QueryEncoding::SameAsStorage, internal queries), 4 or 8 (Scalar4bits / Scalar8bits)
    // TurboQuant (scan_tq.hip): per-row extras columns, bits per value (4 | 2 | 1), VectorParameters.invert
    const float *tq_sf;        // [n] scaling_factor
    const float *tq_l2;        // [n] l2_length (DistanceType::L2) or nullptr
    uint32_t tq_bits, tq_invert;
    uint32_t tq_planes;        // 1-bit storage: bit planes of the query (8; 16 under TQ+)
    uint32_t tq_qbytes_off;    // 1-bit storage: byte offset, inside a query entry, of the i8 form of the query (the matrix-core scan's operand)
    uint32_t tq_i32;           // the integer dot of a (row, query) pair fits 32 bits (scan_sq_mfma.hip finish): 4 / 2 bits below 2000 coordinates, 1 bit below 16384
    uint32_t tq_code_bytes;    // HNSW build (HopTQInternal): code bytes of a row, before the zero padding of the device block
    TqEc tq_ec;                // ... and score_symmetric_ec's inputs (weights == nullptr without TQ+)
    // multi-vectors (MaxSim walk, hnsw.hpp HopMaxSim): point p = inner rows [mv_offsets[p], mv_offsets[p + 1]); multi-query j = query entries
    // [mv_qfirst[j], mv_qfirst[j + 1])
    const uint64_t *mv_offsets;
    uint32_t mv_q_tokens;      // HopMaxSimQ (the build over multi-vector points with PQ / TurboQuant inner rows): query entries of the point being inserted
    const uint32_t *mv_qfirst;
    // custom queries as the walk's scorer (hnsw.hpp HopCustom): search qi = custom query cq_desc[qi] over the example entries of `queries`
    const qmx_custom_query *cq_desc;
    const float *cq_coefs;
    // TurboQuant over Manhattan, the walk (tq_l1_policy.hpp): the segment's TqL1Dev (tq_rotate.hpp) on the device
    const void *tq_l1;
};

enum ScanMode { SCAN_TOPK = 0, SCAN_SCORES = 1 };

// per-query aux block (64 bytes after the padded elements of each query in the tile)
// Byte form of a 1-bit-storage query (scan_sq_mfma.hip Tq1Ops<1> / BqOps): row piece p = 4 s + kg of step s sits at slot 4 s + (0, 2, 1, 3)[kg]
__host__ __device__ inline uint32_t byte_form_slot(uint32_t piece) { return (piece & ~3u) | ((piece & 1u) << 1) | ((piece >> 1) & 1u); }

struct QueryAux {
    float f0;           // u8 cosine: norm1 (reference order) ; SQ: query offset
    int32_t i0;         // u8 cosine scalar order: norm1 as i32
    uint32_t pad[14];
};
constexpr uint32_t QUERY_AUX_BYTES = 64;

// Device-resident HNSW search (hnsw.hpp): the graph + per-launch parameters.
struct HnswArgs {
    const uint32_t *reindex;        // [n_points]   point -> slot inside levels > 0
    const uint64_t *level_offsets;  // [n_levels + 1] first offsets-slot of each level
    const uint64_t *offsets;        // [n_slots + 1]  start of each links list inside `neighbors`
    const uint32_t *neighbors;
    uint64_t n_offsets, n_neighbors; // entries of `offsets` / `neighbors`: the walk never reads past them, whatever a links file claims
    const uint32_t *l0;             // optional packed level 0: l0[p * l0_stride] = count, then the links (one round trip per hop)
    uint32_t l0_stride;
    uint32_t l0_aux_off;            // 0, or m0: l0 is then the wider table of an SQ graph - a row = [m0 link slots, 0xFFFFFFFF behind the last][m0 f32: the linked rows' vector_offset]
    uint32_t n_points, n_levels, m, m0;
    const uint32_t *ep_ids, *ep_levels;   // EntryPoints::entry_points
    uint32_t n_ep;
    const uint32_t *xp_ids, *xp_levels;   // EntryPoints::extra_entry_points (iter_unsorted order)
    uint32_t n_xp;
    uint32_t ef, top, nq;
    uint32_t *visited;              // [slots][vis_words], all zero between searches
    uint64_t vis_words;
    uint32_t *vis_log;              // [slots][log_cap] word indices touched by the running search
    uint32_t log_cap;
    qmx_scored_point *out;          // [nq][top]
    uint32_t *out_counts;           // [nq]
    uint32_t *out_scored;           // [nq] points scored by each search (HardwareCounter cpu_io), may be null
    unsigned long long *pq_stats;   // PQ walk with the hop prefilter: [0] += hop candidates that met the 8-bit bound, [1] += those scored exactly (the survivors); may be null
    uint32_t lds_query_bytes;       // bytes of the query entry staged in LDS (16-byte multiple)
    uint32_t acorn;                 // SearchAlgorithm::Acorn on level 0 (graph_layers.rs:154-243): `visited` holds two bitmaps of vis_words / 2 words
    uint32_t hop_cap;               // entries of the hop id / score buffers in LDS (64; m0 (m0 + 1) rounded up for ACORN)
    // search_on_level_with_vectors (graph_layers.rs:336-389): the candidates the level-0 loop POPS (the one that ends it included) are what
    // the base scorer sees; they are listed here for the base scoring that follows the walk (api_hnsw.hip qmx_hnsw_search_with_vectors)
    uint32_t *expanded;             // [nq][xcap] or nullptr
    uint32_t *expanded_cnt;         // [nq] popped candidates (may exceed xcap: the list is then incomplete)
    uint32_t xcap;
    uint64_t *ev_spill;             // search_with_vectors: [slots][ev_cap] keys - the evicted-unexpanded candidates of the latest score beyond the four in registers
    uint32_t ev_cap;
    // the plain walk's pop sequence (qmx_hnsw_search_traced): every candidate the level-0 loop pops AND expands, in order, with its score
    qmx_scored_point *pops;         // [nq][pop_cap] or nullptr
    uint32_t *pop_cnt;              // [nq] (may exceed pop_cap: the list is then incomplete)
    uint32_t pop_cap;
    // option hnsw_reference_heap_order (a verification mode): `nearest` and `candidates` are the reference's two binary heaps, worked by one lane
    // in std's exact sift order; `nearest` lives in LDS, `candidates` (unbounded in the reference) in this per-slot scratch
    // PQ walk (HopPQ): per search an 8-bit image of its LUT (pq.hip pq_walk_lut8_kernel: [32-byte header: L, Es, step as f64, usable][m x 256 bytes]) staged in
    // LDS; a hop's candidates are scored against it first and only those whose upper bound reaches the beam's worst score are scored exactly
    const unsigned char *pq8;       // [nq][pq8_stride] or nullptr
    uint32_t pq8_stride;            // bytes per search (16-byte multiple) = what is staged in LDS
    uint32_t *next_query;           // device counter the slots draw their next search from (starts at the grid size); nullptr: the static stride
    uint32_t vis_lds;               // bytes of the search's visited table in LDS (hnsw.hpp LdsVisited: 0 or HNSW_VIS_LDS_BYTES); the HBM bitmap then holds what its buckets cannot
    uint32_t ref_heaps;
    uint32_t ref_cap;               // entries of a slot's candidates heap; a search that needs more raises err_flag = 2
    uint2 *ref_cands;               // [slots][ref_cap]  (x = idx, y = score bits)
};

// The PQ walk without LUTs (pq.hip HopPQDirect): the query entry is the preprocessed f32 vector, a LUT entry is recomputed from the codebook where it is needed
bool pq_direct_walk_ok(uint32_t dim, uint32_t m, uint32_t chunk, uint32_t ncent);
int32_t launch_hnsw_pq_direct(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu);
// grid == 0: only report the occupancy (blocks of one wave per CU) of the instantiation in *per_cu
int32_t launch_hnsw_dense(hipStream_t st, int dtype, int distance, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu);
// TurboQuant (scan_tq.hip)
struct TqRotationHost {        // HadamardRotation on the device: forward maps, chunk decomposition, per-chunk 1 / sqrt(size)
    const uint32_t *d_maps, *d_chunk_off, *d_chunk_size;
    const double *d_chunk_norm;
    uint32_t n_chunks, rot_dim, padded_dim, dim;
};
int32_t launch_scan_tq(hipStream_t st, int qt, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid_out);
int32_t launch_scan_bq_mfma(hipStream_t st, int qt, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid_out);
int32_t launch_scan_tq_mfma(hipStream_t st, int qt, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid_out);   // scan_sq_mfma.hip, 4 / 2 bits
int32_t launch_hnsw_tq(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu);
int32_t launch_hnsw_tq_l1(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu, uint32_t rot_dim);   // hnsw_tq_l1.hip
int32_t launch_tq_split(hipStream_t st, const void *rows, uint64_t src_stride, uint64_t n, uint32_t code_bytes, uint32_t dst_stride, int has_l2,
                        void *codes, float *sf, float *l2, float *xm);
int32_t launch_tq_rotate(hipStream_t st, const float *d_in, uint32_t n, const TqRotationHost &h, double *d_out);
int32_t launch_tq_rotate_f64(hipStream_t st, double *d_buf, uint32_t n, const TqRotationHost &h);   // in place, h = the inverse rotation's tables
// TurboQuant over Manhattan (tq_l1.hip): DistanceType::L1 scores = dequantise + inverse rotation per row, then sum |q - v|
struct PairSel;
int32_t launch_tq_l1_dequant(hipStream_t st, const void *codes, uint64_t row_stride, const float *sf, const uint32_t *d_ids, uint64_t id0, uint64_t n,
                             uint64_t n_rows, uint32_t padded_dim, uint32_t value_bits, const float *d_shift, const float *d_scale, double *d_out, int *err_flag,
                             const PairSel *sel);   // rows d_ids[r], or id0 + r; sel: item id0 + r of a pair list (dead slots give zeros)
int32_t launch_tq_l1_diff(hipStream_t st, double *d_a, const double *d_b, uint64_t n_elems);      // a <- a - b
// scores[q * stride + col0 + i] for rows i of d_deq ([n][padded_dim]) and queries [q0, q0 + nq): the sum over k < dim of (f32)|q[k] - v[k]| in order,
// negated when `invert`; d_queries = f32 [.][q_dim] (nullptr: the zero query); sel != nullptr: one (query, row) pair per row of d_deq, scores[col0 + i]
int32_t launch_tq_l1_scores(hipStream_t st, const double *d_deq, uint64_t n, uint32_t padded_dim, uint32_t dim, const float *d_queries, uint32_t q_dim,
                            uint32_t q0, uint32_t nq, float *d_scores, uint64_t stride, uint64_t col0, int invert, const PairSel *sel, uint64_t item0);
int32_t launch_tq_plus_fit(hipStream_t st, double *d_rot, uint32_t n, uint32_t padded_dim, uint32_t distance, double min_q, double max_q, float c_outer,
                           float *d_shift, float *d_scale);
int32_t launch_tq_quantize(hipStream_t st, double *d_rot, uint32_t n, uint32_t padded_dim, uint32_t value_bits, uint32_t distance, void *d_out,
                           uint32_t out_stride, const float *d_shift, const float *d_scale);
int32_t launch_tq_query_encode(hipStream_t st, double *d_rot, uint32_t nq, uint32_t padded_dim, uint32_t bits, int need_l2, void *tile, uint32_t q_stride,
                               uint32_t aux_off, const float *d_shift, const float *d_scale, uint32_t qbytes_off);
int32_t launch_tq_internal(hipStream_t st, const void *codes, uint32_t stride, const float *sf, const float *l2, uint32_t code_bytes, uint32_t bits,
                           int invert, uint64_t n_rows, const uint32_t *a_ids, const uint32_t *b_ids, uint32_t n, float *out, int *err_flag, const TqEc *ec);
// the MaxSim walk over multi-vector points (HopMaxSim): dense, SQ and BQ inner rows
// ... with a custom query (Recommend / Discover / Context / Feedback) as the scorer (hnsw.hpp HopCustom over the storage's own hop policy)
int32_t launch_hnsw_custom_dense(hipStream_t st, int dtype, int distance, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu);
int32_t launch_hnsw_custom_sq(hipStream_t st, int distance, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu);
int32_t launch_hnsw_custom_bq(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu);
int32_t launch_hnsw_custom_pq(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu);
int32_t launch_hnsw_custom_tq(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu);
int32_t launch_hnsw_custom_tq_l1(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu, uint32_t rot_dim);   // (hnsw_tq_l1.hip)
// ... whose examples are multi-vectors over multi-vector points (MultiCustomQueryScorer: HopCustom over HopMaxSim); dense and SQ inner rows
int32_t launch_hnsw_custom_maxsim_dense(hipStream_t st, int dtype, int distance, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu);
int32_t launch_hnsw_custom_maxsim_sq(hipStream_t st, int distance, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu);
int32_t launch_hnsw_maxsim_dense(hipStream_t st, int dtype, int distance, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu);
int32_t launch_hnsw_maxsim_sq(hipStream_t st, int distance, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu);
int32_t launch_hnsw_maxsim_bq(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu);
int32_t launch_hnsw_maxsim_pq(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu);      // (pq.hip; the query's tokens = their LUTs)
int32_t launch_hnsw_maxsim_tq(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu);      // (hnsw_maxsim_tq.hip)
int32_t launch_hnsw_sq(hipStream_t st, int distance, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu);
int32_t launch_hnsw_pq(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu);
// the 8-bit images of a batch's LUTs for the walk's hop prefilter (HnswArgs::pq8); bytes per search: pq_walk_lut8_stride(m)
static inline uint32_t pq_walk_lut8_stride(uint32_t m) { return 32u + m * 256u; }
int32_t launch_pq_walk_lut8(hipStream_t st, const void *d_luts, uint32_t q_stride, uint32_t nq, uint32_t m, uint32_t ncent, void *d_out);
int32_t launch_hnsw_bq(hipStream_t st, const ScanArgs &a, const HnswArgs &h, uint32_t grid, int *per_cu);
int32_t launch_hnsw_pack_level0(hipStream_t st, const uint64_t *offsets, const uint32_t *neighbors, uint32_t n_points, uint32_t stride, uint32_t *l0);
// the table again, each row followed by one f32 per link slot: aux[linked row] (SQ: the offsets column travels with the links)
int32_t launch_hnsw_pack_level0_aux(hipStream_t st, const uint32_t *l0, uint32_t n_points, uint32_t stride, const float *aux, uint64_t n_aux, uint32_t *l0x);
constexpr uint32_t HNSW_VIS_LDS_BYTES = 16384;                 // the walk's visited table in LDS (hnsw.hpp LdsVisited): 1024 buckets x 8 tags of 16 bits
constexpr uint32_t HNSW_VIS_LDS_MAX_POINTS = 65534u * 1024u;    // ... graphs whose (id >> 10) + 1 fits a tag below the "taken back" mark 0xFFFF
constexpr uint32_t HNSW_REF_CAND_CAP = 1u << 16;   // option hnsw_reference_heap_order: entries of one search's `candidates` heap (512 KiB per slot)
constexpr uint32_t HNSW_EV_SPILL_CAP = 4096;      // search_with_vectors: evicted candidates of ONE score a slot can hold beyond four (32 KiB per slot; more raises err_flag = 2)
constexpr uint32_t HNSW_REF_SLOT_CAP = 4096;       // ... searches in flight in that mode (1 024 until round 5: fewer than the default walk keeps in flight)
constexpr uint32_t HNSW_REF_CAND_LDS = 1024;       // ... entries of that heap kept in LDS (8 KiB per search): the ten levels every sift touches
constexpr uint32_t HNSW_MAX_EF = 4096;          // max(top, ef) of a walk: up to 512 in a register beam, beyond it in an LDS beam (hnsw.hpp Beam<0>)
constexpr uint32_t HNSW_BUILD_MAX_M0 = 128;     // links per level-0 list of a device build (m <= m0 <= 128)
constexpr uint32_t HNSW_MAX_EF_REG = 512;       // ... and of ef_construct (the build keeps its beam in registers)
// HNSW build (hnsw_build.hpp)
constexpr uint32_t HNSW_BUILD_MAX_LEVELS = 16;   // levels 0..15 (P(level >= 16) ~ m^-15.5)
struct BuildLinks {
    uint32_t *links0, *cnt0, *linksU, *cntU;
    const uint32_t *up_off;
    uint32_t m, m0;
#if defined(__HIPCC__)
    __device__ __forceinline__ uint32_t *list(uint32_t p, uint32_t level) const {
        return level == 0 ? links0 + (uint64_t)p * m0 : linksU + ((uint64_t)up_off[p] + level - 1) * m;
    }
    __device__ __forceinline__ uint32_t *count(uint32_t p, uint32_t level) const {
        return level == 0 ? cnt0 + p : cntU + ((uint64_t)up_off[p] + level - 1);
    }
    __device__ __forceinline__ uint32_t level_m(uint32_t level) const { return level == 0 ? m0 : m; }
#endif
};
struct HnswBuildArgs {
    BuildLinks g;
    const uint8_t *level;        // [n] level of every point
    uint32_t n_points;
    uint32_t first, count;       // the batch: points first .. first + count
    uint32_t ep_id, ep_level;    // entry point of the graph before the batch
    uint32_t ef_construct;
    uint32_t *visited;           // [slots][vis_words]
    uint64_t vis_words;
    uint32_t *vis_log;           // [slots][log_cap]
    uint32_t log_cap;
    // phase 1 -> phase 2: sel[(bi * HNSW_BUILD_MAX_LEVELS + l) * m0 + k]
    uint32_t *sel_ids;
    float *sel_scores;
    uint32_t *sel_cnt;           // [count][HNSW_BUILD_MAX_LEVELS]
    uint32_t *lock;              // [n]
    uint32_t lds_query_bytes;    // row bytes rounded up to whole 128-byte steps, staged per new point
    uint32_t row_bytes;
    // quantized storages that cannot turn a stored row into a query (PQ): the query entries of the batch, made from the ORIGINAL vectors
    // (point_scorer.rs:197-212) before phase 1; entry bi = batch_queries + bi * batch_q_stride (global memory; nullptr = stage the row)
    const unsigned char *batch_queries;
    uint64_t batch_q_stride;
    uint32_t *next;              // [2] device counters the slots of phase 1 / phase 2 draw their next insertion from (each starts at its launch's grid size); nullptr: static stride
};
// phase 1 = insertion searches + heuristic selection, phase 2 = linking; grid == 0: report occupancy only
int32_t launch_hnsw_build_bq(hipStream_t st, const ScanArgs &a, const HnswBuildArgs &h, int phase, uint32_t grid, int *per_cu);
int32_t launch_hnsw_build_sq(hipStream_t st, int distance, const ScanArgs &a, const HnswBuildArgs &h, int phase, uint32_t grid, int *per_cu);
int32_t launch_hnsw_build_dense(hipStream_t st, int dtype, int distance, const ScanArgs &a, const HnswBuildArgs &h, int phase,
                                uint32_t grid, int *per_cu);
int32_t launch_hnsw_build_maxsim_dense(hipStream_t st, int dtype, int distance, const ScanArgs &a, const HnswBuildArgs &h, int phase, uint32_t grid,
                                       int *per_cu);
int32_t launch_hnsw_build_maxsim_sq(hipStream_t st, int distance, const ScanArgs &a, const HnswBuildArgs &h, int phase, uint32_t grid, int *per_cu);
int32_t launch_hnsw_build_maxsim_bq(hipStream_t st, const ScanArgs &a, const HnswBuildArgs &h, int phase, uint32_t grid, int *per_cu);
int32_t launch_hnsw_build_tq(hipStream_t st, const ScanArgs &a, const HnswBuildArgs &h, int phase, uint32_t grid, int *per_cu);
int32_t launch_hnsw_build_tq_l1(hipStream_t st, const ScanArgs &a, const HnswBuildArgs &h, int phase, uint32_t grid, int *per_cu, uint32_t rot_dim,
                                uint32_t padded_dim);      // ... over Manhattan (hnsw_build_tq_l1.hip)
int32_t launch_hnsw_build_pq(hipStream_t st, const ScanArgs &a, const HnswBuildArgs &h, int phase, uint32_t grid, int *per_cu);
int32_t launch_hnsw_build_maxsim_pq(hipStream_t st, const ScanArgs &a, const HnswBuildArgs &h, int phase, uint32_t grid, int *per_cu);   // multi-vector points over PQ inner rows
int32_t launch_hnsw_build_maxsim_tq(hipStream_t st, const ScanArgs &a, const HnswBuildArgs &h, int phase, uint32_t grid, int *per_cu);   // ... over TurboQuant inner rows (hnsw_build_multi_tq.hip)
// pair[c][i][j] = DistanceType::distance(centroid i chunk c, centroid j chunk c): the per-chunk terms of score_internal (encoded_vectors_pq.rs:574-618)
int32_t launch_pq_pair_table(hipStream_t st, uint32_t distance, uint32_t dim, const qmx_pq_params &pq, const float *d_centroids, float *d_pair);
// norms[r] = sum of squares of u8 row r as the cosine leaf computes it for a stored row (metric_uint/avx2/cosine.rs, simple_cosine.rs)
int32_t launch_u8_row_norms(hipStream_t st, const void *rows, uint64_t row_stride, uint64_t n, uint32_t dim, uint32_t flags, float *norms_f, int32_t *norms_i);
constexpr uint32_t HNSW_LDS_QUERY_MAX = 150 * 1024;

// dense f32 / f16 / u8 (scan_dense.hip)
int32_t launch_scan_dense(hipStream_t st, int dtype, int distance, int qt, ScanMode mode,
                          const ScanArgs &a, int num_cus, uint32_t *grid_out);
// f32 dot / cosine, 8..32 queries per pass on v_mfma_f32_4x4x1 with the reference's bits (scan_mfma.hip)
int32_t launch_scan_f32_mfma(hipStream_t st, int qt, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid_out);
// SQ int8 dot / cosine / euclid, 8..32 queries per pass on v_mfma_i32_16x16x64_i8 (scan_sq_mfma.hip)
bool sq_mfma_ok(uint32_t distance, uint32_t actual_dim);
int32_t launch_scan_sq_mfma(hipStream_t st, int qt, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid_out);
int32_t launch_scan_f16_mfma(hipStream_t st, int qt, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid_out);
// which query an item belongs to: explicit list, or fixed-size slots (item / per_query, with the
// live prefix of each slot given by counts), or item == query (score_internal)
struct PairSel {
    const uint32_t *qsel;      // explicit query index per item, or nullptr
    uint32_t per_query;        // > 0: query = item / per_query
    const uint32_t *counts;    // with per_query: only the first counts[query] items of a slot are live
    const uint32_t *limit;     // or nullptr: a device-side item count - items at and behind *limit are dead (a pool filled on the device: the host knows its capacity only)
    __device__ __forceinline__ uint32_t query_of(uint64_t item) const {
        if (qsel) return qsel[item];
        if (per_query) return (uint32_t)(item / per_query);
        return (uint32_t)item;
    }
    __device__ __forceinline__ bool live(uint64_t item, uint32_t qi) const {
        if (limit && item >= (uint64_t)*limit) return false;
        if (per_query && counts) return (uint32_t)(item % per_query) < counts[qi];
        return true;
    }
};

int32_t launch_pairs_dense(hipStream_t st, int dtype, int distance, const ScanArgs &a, const PairSel &sel,
                           uint64_t n_items, int num_cus);
int32_t launch_pairs_sq(hipStream_t st, int distance, const ScanArgs &a, const PairSel &sel, uint64_t n_items, int num_cus);
// SQ int8 rows: codes SoA + offsets (scan_quant.hip)
int32_t launch_scan_sq(hipStream_t st, int distance, int qt, ScanMode mode, const ScanArgs &a, int num_cus,
                       uint32_t *grid_out);
int32_t launch_sq_encode(hipStream_t st, int distance, const qmx_sq_params &sp, uint32_t dim, const float *in, uint64_t n,
                         uint8_t *codes_out, uint64_t codes_stride, float *offsets_out, uint8_t *rows_out, int is_query,
                         uint32_t aux_off);
int32_t launch_sq_split(hipStream_t st, const void *rows, uint64_t row_stride, uint64_t n, uint32_t actual_dim, void *codes,
                        float *offsets);
int32_t launch_sq_gather_rows(hipStream_t st, const void *codes, const float *offsets, uint32_t actual_dim, const uint32_t *ids,
                              uint32_t n, uint64_t n_rows, void *rows_out, int *err_flag);
int32_t launch_sq_internal_query(hipStream_t st, const void *codes, const float *offsets, uint32_t actual_dim, const uint32_t *ids,
                                 uint32_t nq, uint64_t n_rows, float shift, void *tile, uint32_t q_stride, uint32_t aux_off,
                                 int *err_flag);
// f32 dot / cosine, 32- and 64-query tiles on v_mfma_f32_16x16x4_f32, chain-major (scan_mfma16.hip)
bool mfma16_dim_ok(int qt, uint32_t dim);
bool mfma16_scan_ok(int qt, ScanMode mode, const ScanArgs &a);   // qt = 16, 32 or 64
int32_t launch_scan_f32_mfma16(hipStream_t st, int qt, const ScanArgs &a, int num_cus, uint32_t *grid_out);
// f32 dot / cosine, 65..128 queries per pass: f16-split matrix-core prefilter + exact verification (scan_split.hip)
bool split_scan_ok(const ScanArgs &a);
size_t split_query_bytes(uint32_t dim);
float split_row_scale(float row_maxabs);
int32_t launch_split_row_stats(hipStream_t st, const void *rows, uint64_t row_stride, uint64_t n, uint32_t dim, uint32_t *d_stats);
int32_t launch_split_pack_queries(hipStream_t st, const float *d_q, uint32_t nq, uint32_t dim, float *d_qmax, float *d_qnorm, void *d_bq, int half, uint32_t qt);
int32_t launch_split_thresholds(hipStream_t st, const uint64_t *d_gthr, const float *d_qnorm, const float *d_qmax, uint32_t nq, float rel_band, float row_norm_max,
                                float row_scale, float *d_scales, float *d_thr, float *d_band, uint32_t qt, uint32_t *d_cand_cnt, uint32_t n_cnt);
int32_t launch_scan_f32_split(hipStream_t st, const ScanArgs &a, const void *d_bq, float row_scale, const float *d_scales, const float *d_thr,
                              uint64_t *d_cand, uint32_t *d_cand_cnt, uint32_t cap, int num_cus, const void *d_rows_split, int half, void *d_wlists, uint32_t phase, uint32_t qt);
int32_t launch_regroup_lists(hipStream_t st, const DeletedView &del, const void *d_wlist, const uint32_t *d_wcnt, uint32_t wcap, uint32_t n_lists, uint64_t *d_cand,
                             uint32_t *d_cand_cnt, uint32_t cap, int *d_overflow);
// PQ prefilter (pq_prefilter.hip): rotated copy of the code block, 6-bit tables + thresholds per query, the approximate scan
size_t pq_rot_bytes(uint64_t n, uint32_t m);
bool pq_prefilter_shape_ok(uint32_t m, uint32_t ncent);
int32_t launch_pq_rotate(hipStream_t st, const void *codes, uint64_t row_stride, uint64_t n, uint32_t m, void *d_out);
size_t pq_prefilter_table_bytes(uint32_t m, uint32_t nq);
uint32_t pq_prefilter_grid(int num_cus, uint32_t nq, uint32_t *n_slabs_out);
size_t pq_prefilter_wlists_counts_bytes(uint32_t grid);
size_t pq_prefilter_wlists_bytes(uint32_t grid, uint32_t wcap);
int32_t launch_pq_lut8(hipStream_t st, const void *d_luts, uint32_t q_stride, uint32_t nq, uint32_t m, uint32_t ncent, const uint64_t *d_gthr, void *d_table8,
                       int32_t *d_thr, float *d_band);
int32_t launch_pq_prefilter(hipStream_t st, const ScanArgs &a, const void *d_rot, const void *d_table8, const int32_t *d_thr, uint32_t nq, int num_cus,
                            void *d_wlists, uint32_t wcap, uint32_t *grid_out);
int32_t launch_split_refine(hipStream_t st, const uint64_t *d_cand, const uint32_t *d_cand_cnt, uint32_t cap, const float *d_band, uint32_t nq, uint32_t top,
                            const float *d_scales, float *d_thr);
size_t split_wlists_bytes(int num_cus);
int32_t launch_split_regroup(hipStream_t st, const ScanArgs &a, const void *d_wlists, int num_cus, uint64_t *d_cand, uint32_t *d_cand_cnt, uint32_t cap,
                             int *d_overflow, uint32_t phase, uint32_t qt);
size_t split_copy_bytes(uint64_t n, uint32_t dim, int half);
int32_t launch_split_copy(hipStream_t st, const void *rows, uint64_t row_stride, uint64_t n, uint32_t dim, float row_scale, void *d_out, int half);
// what one search through the prefilter did (device side; api_search.hip folds it into qmx_counters)
struct SplitStats {
    unsigned long long candidates;   // (row, query) pairs the approximate scan let through, deleted rows dropped
    unsigned long long verified;     // of those, re-scored exactly from the f32 rows
    uint32_t fallback_queries;       // queries whose lists overflowed: they took the exact scan of the block
    uint32_t pad;
};
// the verification pool of one search: the rows worth an exact score, of all its queries, as one ragged list filled on the device - query q's rows sit at
// ids[off[q] .. off[q] + cnt[q]) with qsel[] = q beside them (the gather kernel's explicit query index), *used = entries taken so far (zeroed per search).
// A query takes what it needs (a few hundred rows on friendly data, tens of thousands where a worst-case band is wide): only a pool that is full sends a
// query to the exact scan.
struct VerifyPool {
    uint32_t *ids, *qsel, *off, *cnt, *used;
    uint32_t cap;
    uint32_t max_per_query;     // a query with more rows than this inside its band takes the exact scan whatever room the pool has (option verify_max_per_query; default: cap)
};
int32_t launch_split_select(hipStream_t st, const uint64_t *d_cand, const uint32_t *d_cand_cnt, uint32_t cap, const float *d_band, uint32_t nq, uint32_t top,
                            const VerifyPool &pool, uint32_t q_base, const int *d_tile_overflow, uint32_t *d_ovf_q, SplitStats *d_stats,
                            const float *d_t_exact = nullptr, bool tighten = false);
// the int8 copy of an f32 block (QMX_SEG_I8_COPY) and its passes (scan_split.hip, "The INT8 copy")
bool split_i8_dim_ok(uint32_t dim);
size_t split_i8_copy_bytes(uint64_t n, uint32_t dim);
size_t split_i8_query_bytes(uint32_t dim);
uint32_t split_i8_probe();
int32_t launch_split_i8_colstats(hipStream_t st, const void *rows, uint64_t row_stride, uint64_t n, uint32_t dim, uint32_t *d_colmax, float *d_colsq);
float split_i8_choose_scales(const float *colmax, const float *colsq, uint64_t n, uint32_t dim, float *scale);   // host; returns the balance G (0 = floor scales)
int32_t launch_split_i8_rowstats(hipStream_t st, const void *rows, uint64_t row_stride, uint64_t n, uint32_t dim, const float *d_scale, uint32_t *d_stats);
int32_t launch_split_i8_copy(hipStream_t st, const void *rows, uint64_t row_stride, uint64_t n, uint32_t dim, const float *d_scale, void *d_out);
int32_t launch_split_i8_pack(hipStream_t st, const float *d_q, uint32_t nq, uint32_t dim, const float *d_scale, const uint64_t *d_gthr, const uint32_t *d_row_stats,
                             float row_norm_max, void *d_bq, float *d_qscale, float *d_band, float *d_thr, float *d_t_exact, uint32_t *d_cand_cnt, uint32_t n_cnt);
int32_t launch_scan_i8copy(hipStream_t st, const ScanArgs &a, const void *d_bq, const float *d_qscale, const float *d_thr, int num_cus, const void *d_rows_i8,
                           void *d_wlists, uint32_t phase);
int32_t launch_split_i8_probe(hipStream_t st, const uint64_t *d_cand, const uint32_t *d_cand_cnt, uint32_t cap, const float *d_band, uint32_t nq, uint32_t top,
                              const int *d_tile_overflow, uint32_t *d_probe_ids, uint32_t *d_probe_cnt);
int32_t launch_split_i8_bound(hipStream_t st, const float *d_scores, uint32_t *d_probe_cnt, uint32_t nq, uint32_t top, const float *d_band, const float *d_qscale,
                              float *d_thr, float *d_t_exact);
// TurboQuant 4 bits, 128 queries per pass: codes decoded once per tile into the matrix cores' operand images, exact scores, candidate lists (scan_tq4w.hip)
bool tq4w_shape_ok(const ScanArgs &a);
size_t tq4w_query_bytes(uint32_t code_bytes);
size_t tq4w_wlists_counts_bytes(int num_cus);
size_t tq4w_wlists_bytes(int num_cus);
uint32_t tq4w_wcap();
int32_t launch_tq4w_stats(hipStream_t st, const float *d_sf, const float *d_l2, const void *d_rows, uint64_t row_stride, uint32_t code_bytes, uint64_t n, uint32_t *d_stats);
int32_t launch_tq4w_pack(hipStream_t st, const ScanArgs &a, const uint64_t *d_gthr, float sf_min, float sf_max, float l2_min, float l2_max, uint32_t c1, int high_only,
                         void *d_bq, int32_t *d_thr_i,
                         float *d_qinfo, float *d_band, uint32_t *d_cand_cnt, uint32_t n_cnt);
int32_t launch_scan_tq4w(hipStream_t st, const ScanArgs &a, const void *d_bq, const int32_t *d_thr_i, const float *d_qinfo, const float *d_band_high_only, int num_cus,
                         void *d_wlists, uint32_t *grid_out);
// scalar int8, 128 queries per pass: a wave's lanes fetch their own operand pieces, exact integer dots, candidate lists (scan_sqw.hip)
bool sqw_shape_ok(const ScanArgs &a);
size_t sqw_query_bytes(uint32_t dim);
size_t sqw_wlists_counts_bytes(int num_cus);
size_t sqw_wlists_bytes(int num_cus);
uint32_t sqw_wcap();
int32_t launch_sqw_stats(hipStream_t st, const float *d_off, uint64_t n, float multiplier, int32_t *d_bi, uint32_t *d_stats);
int32_t launch_sqw_pack(hipStream_t st, const ScanArgs &a, const uint64_t *d_gthr, float off_absmax, void *d_bq, int32_t *d_thr_i, float *d_qinfo, float *d_band,
                        uint32_t *d_cand_cnt, uint32_t n_cnt);
int32_t launch_scan_sqw(hipStream_t st, const ScanArgs &a, const void *d_bq, const int32_t *d_thr_i, const int32_t *d_bi, const float *d_qinfo, int num_cus,
                        void *d_wlists, uint32_t *grid_out);
// the overflowed queries packed for the conditional exact passes: list, their pre-scan bounds, the passes' run flags
int32_t launch_split_plan(hipStream_t st, const uint32_t *d_ovf_q, uint32_t nq, const uint64_t *d_gthr, uint32_t *d_list, uint64_t *d_gthr_packed, uint32_t list_cap,
                          uint32_t *d_count, int *d_run16, int *d_run64, uint32_t n_run64, SplitStats *d_stats, const void *d_queries, uint32_t q_stride,
                          void *d_packed_queries);
// order statistics of a float array (quantile.hip): the SQ quantile interval
int32_t launch_order_statistics_f32(hipStream_t st, const float *d_in, float *d_tmp, uint64_t n, uint64_t lo_pos, uint64_t hi_pos, float *h_out);
// BQ 1-bit (scan_bq.hip)
int32_t launch_scan_bq(hipStream_t st, int qt, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid_out);
int32_t launch_pairs_bq(hipStream_t st, const ScanArgs &a, const PairSel &sel, uint64_t n_items, int num_cus);
int32_t launch_pairs_tq(hipStream_t st, const ScanArgs &a, const PairSel &sel, uint64_t n_items, int num_cus);
uint64_t bq_row_bytes(uint32_t dim, uint32_t encoding);
int32_t launch_bq_encode_scalar_query(hipStream_t st, const float *d_in, uint32_t nq, uint32_t dim, uint32_t encoding, uint32_t bits, uint8_t *d_out,
                                      uint32_t out_stride, uint32_t qbytes_off, uint32_t aux_off, uint32_t body);
int32_t launch_vector_stats(hipStream_t st, const float *d_rows, uint64_t row_stride_bytes, uint64_t n, uint32_t dim, float *d_min, float *d_max, float *d_mean,
                            float *d_stddev);
int32_t launch_bq_encode(hipStream_t st, const float *d_in, uint64_t n, uint32_t dim, uint32_t encoding, const float *d_mean, const float *d_stddev,
                         uint8_t *d_out, uint64_t out_stride);
// PQ (pq.hip)
int32_t launch_scan_pq(hipStream_t st, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid_out);
int32_t launch_pairs_pq(hipStream_t st, const ScanArgs &a, const PairSel &sel, uint64_t n_items, int num_cus);
int32_t launch_pq_lut(hipStream_t st, uint32_t distance, uint32_t dim, const qmx_pq_params &pq, const float *d_centroids,
                      const float *d_queries, uint32_t nq, float *d_lut);
int32_t launch_pq_internal(hipStream_t st, uint32_t distance, uint32_t dim, const qmx_pq_params &pq, const float *d_centroids,
                           const float *d_pair, const void *rows, uint64_t row_stride, uint64_t n_rows, const uint32_t *a_ids, const uint32_t *b_ids,
                           uint32_t n, float *out, int *err_flag);
int32_t launch_pq_train(hipStream_t st, uint32_t dim, uint32_t chunk_size, uint32_t n_centroids, const float *d_data, uint64_t n,
                        uint32_t max_iters, float accuracy, uint32_t threads, float *d_centroids, uint32_t *iters_host);
int32_t launch_pq_encode(hipStream_t st, uint32_t dim, const qmx_pq_params &pq, const float *d_centroids, const float *d_in,
                         uint64_t n, uint8_t *d_codes);
// query tile packing: preprocessed f32 queries -> element type, padded, + aux (preprocess.hip)
int32_t launch_pack_queries(hipStream_t st, int dtype, int distance, const void *src, int src_is_encoded,
                            uint32_t src_stride, uint32_t nq, uint32_t dim, void *tile, uint32_t q_stride, uint32_t aux_off);
// top-k merge of `n_lists` key lists per query into ScoredPointOffset rows (topk_merge.hip)
// pass p of a top > 64 search: writes out[q * out_stride + out_offset ..+top), counts accumulate when out_offset > 0,
// next_bound[q] = key of the last entry written (0 = this query is exhausted)
int32_t launch_merge_keys(hipStream_t st, const uint64_t *partial, uint32_t n_lists,
                          uint32_t qt_stride, uint32_t nq, uint32_t top, qmx_scored_point *out,
                          uint32_t *out_counts, uint32_t out_stride = 0, uint32_t out_offset = 0,
                          uint64_t *next_bound = nullptr, const int *run_if = nullptr, const uint32_t *out_map = nullptr, uint32_t shared_grid = 0);
int32_t launch_merge_points(hipStream_t st, const qmx_scored_point *lists, const uint32_t *list_counts,
                            const uint32_t *list_idx_base, uint32_t n_lists, uint32_t nq, uint32_t k, qmx_scored_point *out,
                            uint32_t *out_counts, uint64_t list_stride = 0, uint64_t count_stride = 0);   // strides in entries / words between lists (0 = contiguous arrays)
int32_t launch_split_candidates(hipStream_t st, const qmx_scored_point *cand, const uint32_t *cand_cnt, uint32_t n_per, uint32_t nq,
                                uint32_t *ids, uint32_t top, qmx_scored_point *out, uint32_t *out_counts);
int32_t launch_sort_scored(hipStream_t st, const float *scores, const uint32_t *ids, const uint32_t *counts, uint32_t n_per_query, uint32_t nq, uint32_t top,
                           qmx_scored_point *out, uint32_t *out_counts, const uint32_t *offsets = nullptr);   // offsets: query q's entries start at offsets[q] (ragged), not at q * n_per_query

// custom queries (custom_query.hip): combine the per-example similarity matrix, top-k of a score row
int32_t launch_custom_combine(hipStream_t st, const qmx_custom_query *d_queries, uint32_t n_queries, const float *d_sims, uint64_t n,
                              const float *d_coefs, float *d_out);
int32_t launch_maxsim(hipStream_t st, const float *d_sims, uint64_t n_rows, const uint32_t *d_qfirst, uint32_t n_queries, const uint64_t *d_offsets,
                      uint32_t n_points, const uint32_t *d_ids, uint64_t n, float *d_out, int *err_flag);
int32_t launch_custom_topk(hipStream_t st, const float *d_scores, uint64_t n, const uint32_t *d_ids, const DeletedView &del, uint32_t n_queries,
                           uint32_t top, qmx_scored_point *d_out, uint32_t *d_counts, uint64_t *d_bound = nullptr);   // d_bound[q] = the k-th best key of a full list, else 0

// Metric::preprocess + element casts (preprocess.hip)
int32_t launch_cosine_preprocess_f32(hipStream_t st, const float *in, float *out, uint64_t n, uint32_t dim);
int32_t launch_cast_f32(hipStream_t st, int dst_dtype, const float *in, void *out, uint64_t count);
int32_t launch_minmax_f32(hipStream_t st, const float *in, uint64_t count, float *min_out, float *max_out);
int32_t launch_gather_rows(hipStream_t st, const void *rows, uint64_t row_stride, uint64_t row_bytes,
                           const uint32_t *ids, uint32_t n, uint64_t n_rows, void *out, int *err_flag);

}  // namespace qmx

/* qdrant_oracle_links.c — CPU restatement of the reference's COMPRESSED graph-links file formats.
 *
 * TEST INFRASTRUCTURE ONLY (see qdrant_oracle.h): it writes the files the library's reader
 * (qdrant_amd/csrc/links_file.hip) is checked against, and re-reads them with the reference's own
 * iterator logic.  Nothing under qdrant_amd/ links, imports or executes this file.
 *
 * Follows, function by function:
 *   lib/common/common/src/bitpacking.rs          :11-57   BitWriter::{write, finish}
 *                                                :60-152  BitReader::{set_bits, read}, read_buf_and_advance
 *                                                :154-166 packed_bits, make_bitmask
 *   lib/common/common/src/bitpacking_links.rs    :38-82   pack_links
 *                                                :90-119  iterate_packed_links
 *                                                :124-151 packed_links_size
 *                                                :153-207 PackedLinksIterator::{next_sorted, next_unsorted, next}
 *   lib/common/common/src/bitpacking_ordered.rs  :69-105  compress / compress_with_parameters
 *                                                :184-229 Parameters::{compressed_size_bytes, chunk_size_bytes, find_best, try_all}
 *                                                :303-316 Reader::decode_chunk
 *   lib/segment/src/index/hnsw_index/graph_links/serializer.rs :23-243 serialize_graph_links (Compressed, CompressedWithVectors)
 *   lib/segment/src/index/hnsw_index/graph_links/header.rs     :22-54  HeaderCompressed, HeaderCompressedWithVectors
 *   integer-encoding 4.x (Cargo.lock; absent from /root/reference): VarInt for u64 = unsigned LEB128
 *
 * Parity pin: the reference holds no golden BYTES for these formats (its tests are round trips on
 * StdRng data); the only known-answer vector is bitpacking.rs `test_simple` (10 packed bytes and the
 * values read back), checked in tests/test_oracle_links.py.  Beyond that the writer here and the
 * reader here restate the two sides of the reference separately, and the library's reader is an
 * independent third implementation: all three must agree.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "qdrant_oracle.h"

/* ---- growable byte vector (Vec<u8>) ----------------------------------------------------------- */
typedef struct {
    uint8_t *p;
    size_t len, cap;
} bytevec;

static void bv_reserve(bytevec *v, size_t extra) {
    if (v->len + extra <= v->cap) return;
    size_t nc = v->cap ? v->cap * 2 : 256;
    while (nc < v->len + extra) nc *= 2;
    v->p = (uint8_t *)realloc(v->p, nc);
    v->cap = nc;
}
static void bv_push(bytevec *v, const void *src, size_t n) {
    bv_reserve(v, n);
    memcpy(v->p + v->len, src, n);
    v->len += n;
}
static void bv_zeros(bytevec *v, size_t n) {
    bv_reserve(v, n);
    memset(v->p + v->len, 0, n);
    v->len += n;
}

/* ---- bitpacking.rs ------------------------------------------------------------------------------ */
static uint8_t packed_bits_u64(uint64_t max_value) { return max_value ? (uint8_t)(64 - __builtin_clzll(max_value)) : 0; }
static uint8_t packed_bits_u32(uint32_t max_value) { return max_value ? (uint8_t)(32 - __builtin_clz(max_value)) : 0; }
static uint64_t bitmask_u64(uint8_t bits) { return bits >= 64 ? ~0ull : (1ull << bits) - 1ull; }

typedef struct {
    bytevec *out;
    uint64_t buf;
    uint8_t buf_bits;
} bitwriter;

static void bw_init(bitwriter *w, bytevec *out) {
    w->out = out;
    w->buf = 0;
    w->buf_bits = 0;
}
/* BitWriter::write (:24-44) */
static void bw_write(bitwriter *w, uint64_t value, uint8_t bits) {
    w->buf |= value << w->buf_bits;
    w->buf_bits = (uint8_t)(w->buf_bits + bits);
    if (w->buf_bits >= 64) {
        bv_push(w->out, &w->buf, 8);   /* little-endian host */
        w->buf_bits = (uint8_t)(w->buf_bits - 64);
        if ((uint8_t)(bits - w->buf_bits) == 64) w->buf = 0;
        else w->buf = value >> (bits - w->buf_bits);
    }
}
/* BitWriter::finish (:51-56) */
static void bw_finish(bitwriter *w) { bv_push(w->out, &w->buf, ((size_t)w->buf_bits + 7) / 8); }

typedef struct {
    const uint8_t *in;
    size_t in_len;
    uint64_t buf, mask;
    uint8_t buf_bits, bits;
} bitreader;

static void br_init(bitreader *r, const uint8_t *in, size_t len) {
    memset(r, 0, sizeof(*r));
    r->in = in;
    r->in_len = len;
}
static void br_set_bits(bitreader *r, uint8_t bits) {
    r->bits = bits;
    r->mask = bitmask_u64(bits);
}
/* read_buf_and_advance (:133-152) */
static uint64_t br_fetch(bitreader *r) {
    uint64_t b = 0;
    if (r->in_len >= 8) {
        memcpy(&b, r->in, 8);
        r->in += 8;
        r->in_len -= 8;
    } else {
        for (size_t i = 0; i < r->in_len; i++) b |= (uint64_t)r->in[i] << (8 * i);
        /* the reference leaves `input` in place here; nothing reads past it in well-formed data */
    }
    return b;
}
/* BitReader::read (:101-130) */
static uint64_t br_read(bitreader *r) {
    if (r->buf_bits >= r->bits) {
        r->buf_bits = (uint8_t)(r->buf_bits - r->bits);
        const uint64_t val = r->buf & r->mask;
        r->buf = r->bits >= 64 ? 0 : r->buf >> r->bits;
        return val;
    }
    const uint64_t nb = br_fetch(r);
    const uint64_t val = (r->buf | (r->buf_bits >= 64 ? 0 : nb << r->buf_bits)) & r->mask;
    r->buf_bits = (uint8_t)(r->buf_bits + 64 - r->bits);
    r->buf = r->buf_bits == 0 ? 0 : nb >> (64 - r->buf_bits);
    return val;
}

/* known-answer entry points for bitpacking.rs `test_simple` (:181-209): pack (value, bits) pairs, read them back */
uint64_t qo_bitpack_write(const uint64_t *values, const uint8_t *bits, uint32_t n, uint8_t *out, uint64_t cap) {
    bytevec v = {0};
    bitwriter w;
    bw_init(&w, &v);
    for (uint32_t i = 0; i < n; i++) bw_write(&w, values[i], bits[i]);
    bw_finish(&w);
    const uint64_t len = v.len;
    if (out && len <= cap) memcpy(out, v.p, len);
    free(v.p);
    return len;
}
void qo_bitpack_read(const uint8_t *data, uint64_t len, const uint8_t *bits, uint32_t n, uint64_t *values) {
    bitreader r;
    br_init(&r, data, len);
    for (uint32_t i = 0; i < n; i++) {
        br_set_bits(&r, bits[i]);
        values[i] = br_read(&r);
    }
}

/* ---- bitpacking_links.rs -------------------------------------------------------------------------- */
#define MIN_BITS_PER_VALUE 8
#define HEADER_BITS 5

static int cmp_u32(const void *a, const void *b) {
    const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

/* pack_links (:38-82): sorts raw_links[..sorted_count] in place (as the reference does) */
static void pack_links(bytevec *links, uint32_t *raw, size_t n, uint8_t bits_per_unsorted, size_t sorted_count) {
    if (n == 0) return;
    if (sorted_count > n) sorted_count = n;
    qsort(raw, sorted_count, sizeof(uint32_t), cmp_u32);
    for (size_t i = sorted_count; i-- > 1;) raw[i] -= raw[i - 1];
    bitwriter w;
    bw_init(&w, links);
    if (sorted_count != 0) {
        uint32_t mx = 0;
        for (size_t i = 0; i < sorted_count; i++) mx = raw[i] > mx ? raw[i] : mx;
        uint8_t bps = packed_bits_u32(mx);
        if (bps < MIN_BITS_PER_VALUE) bps = MIN_BITS_PER_VALUE;
        bw_write(&w, (uint64_t)(bps - MIN_BITS_PER_VALUE), HEADER_BITS);
        for (size_t i = 0; i < sorted_count; i++) bw_write(&w, raw[i], bps);
    }
    for (size_t i = sorted_count; i < n; i++) bw_write(&w, raw[i], bits_per_unsorted);
    bw_finish(&w);
    for (size_t i = 1; i < sorted_count; i++) raw[i] += raw[i - 1];
}

uint64_t qo_pack_links(uint32_t *raw_links, uint32_t n, uint8_t bits_per_unsorted, uint32_t sorted_count, uint8_t *out, uint64_t cap) {
    bytevec v = {0};
    pack_links(&v, raw_links, n, bits_per_unsorted, sorted_count);
    const uint64_t len = v.len;
    if (out && len <= cap) memcpy(out, v.p, len);
    free(v.p);
    return len;
}

/* iterate_packed_links + PackedLinksIterator::next (:90-119, :161-207), collected; returns the count */
uint32_t qo_iterate_packed_links(const uint8_t *links, uint64_t len, uint8_t bits_per_unsorted, uint32_t sorted_count, uint32_t *out,
                                 uint32_t cap) {
    bitreader r;
    br_init(&r, links, len);
    size_t remaining_bits = (size_t)len * 8, target = remaining_bits;
    if (sorted_count != 0 && len != 0) {
        br_set_bits(&r, HEADER_BITS);
        const uint8_t bps = (uint8_t)(br_read(&r) + MIN_BITS_PER_VALUE);
        remaining_bits -= HEADER_BITS;
        br_set_bits(&r, bps);
        const size_t max_sorted = remaining_bits / bps;
        target -= (sorted_count < max_sorted ? sorted_count : max_sorted) * (size_t)bps;
    } else {
        br_set_bits(&r, bits_per_unsorted);
    }
    uint32_t n = 0, delta = 0;
    for (;;) {
        uint32_t value;
        if (remaining_bits > target) {
            delta += (uint32_t)br_read(&r);   /* wrapping_add */
            remaining_bits -= r.bits;
            value = delta;
            if (remaining_bits <= target) br_set_bits(&r, bits_per_unsorted);
        } else {
            if (remaining_bits < r.bits) break;
            remaining_bits -= r.bits;
            value = (uint32_t)br_read(&r);
        }
        if (n < cap) out[n] = value;
        n++;
    }
    return n;
}

/* packed_links_size (:124-151) */
uint64_t qo_packed_links_size(const uint8_t *data, uint64_t len, uint8_t bits_per_unsorted, uint32_t sorted_count, uint32_t total_count) {
    if (total_count == 0 || len == 0) return 0;
    size_t total_bits = 0;
    const uint32_t actual_sorted = total_count < sorted_count ? total_count : sorted_count;
    if (actual_sorted > 0) {
        total_bits += HEADER_BITS;
        const uint8_t bps = (uint8_t)((data[0] & ((1u << HEADER_BITS) - 1u)) + MIN_BITS_PER_VALUE);
        total_bits += (size_t)actual_sorted * bps;
    }
    total_bits += (size_t)(total_count - actual_sorted) * bits_per_unsorted;
    return (total_bits + 7) / 8;
}

/* ---- bitpacking_ordered.rs ---------------------------------------------------------------------- */
#define TAIL_SIZE 7
#define MAX_CHUNK_LEN_LOG2 7

typedef struct {
    uint64_t length;
    uint8_t base_bits, delta_bits, chunk_len_log2;
} ordered_params;

static size_t ordered_chunk_size(const ordered_params *p) {
    const size_t bits = (size_t)p->base_bits + (size_t)p->delta_bits * (((size_t)1 << p->chunk_len_log2) - 1);
    return (bits + 7) / 8;
}
static size_t ordered_compressed_size(const ordered_params *p) {
    const size_t chunk_len = (size_t)1 << p->chunk_len_log2;
    const size_t chunks = ((size_t)p->length + chunk_len - 1) / chunk_len;
    return chunks * ordered_chunk_size(p) + TAIL_SIZE;
}
/* Parameters::find_best over try_all (:208-229): min_by_key keeps the FIRST minimum */
static ordered_params ordered_find_best(const uint64_t *values, size_t n) {
    ordered_params best;
    memset(&best, 0, sizeof(best));
    int have = 0;
    size_t best_size = 0;
    const uint64_t last = n ? values[n - 1] : 0;
    for (uint8_t cl = 0; cl <= MAX_CHUNK_LEN_LOG2; cl++) {
        uint8_t delta_bits = 1;
        const size_t chunk_len = (size_t)1 << cl;
        for (size_t s = 0; s < n; s += chunk_len) {
            const size_t e = s + chunk_len < n ? s + chunk_len : n;
            const uint8_t pb = packed_bits_u64(values[e - 1] - values[s]);
            if (pb > delta_bits) delta_bits = pb;
        }
        if (delta_bits > 56) continue;
        ordered_params p;
        p.length = n;
        p.base_bits = packed_bits_u64(last);
        if (p.base_bits < 1) p.base_bits = 1;
        p.delta_bits = delta_bits;
        p.chunk_len_log2 = cl;
        const size_t sz = ordered_compressed_size(&p);
        if (!have || sz < best_size) {
            best = p;
            best_size = sz;
            have = 1;
        }
    }
    return best;
}
/* compress_with_parameters (:77-105) */
static void ordered_compress(bytevec *out, const uint64_t *values, size_t n, const ordered_params *p) {
    const size_t chunk_len = (size_t)1 << p->chunk_len_log2;
    for (size_t s = 0; s < n; s += chunk_len) {
        const size_t e = s + chunk_len < n ? s + chunk_len : n;
        bitwriter w;
        bw_init(&w, out);
        bw_write(&w, values[s], p->base_bits);
        for (size_t i = s + 1; i < e; i++) bw_write(&w, values[i] - values[s], p->delta_bits);
        for (size_t i = 0; i < chunk_len - (e - s); i++) bw_write(&w, bitmask_u64(p->delta_bits), p->delta_bits);
        bw_finish(&w);
    }
    for (int i = 0; i < TAIL_SIZE; i++) {
        const uint8_t ff = 0xFF;
        bv_push(out, &ff, 1);
    }
}
/* Reader::decode_chunk (:303-316) through chunk_offset (:296-298) */
static uint64_t ordered_get(const uint8_t *data, const ordered_params *p, size_t index) {
    const uint8_t *chunk = data + (index >> p->chunk_len_log2) * ordered_chunk_size(p);
    uint64_t w0;
    memcpy(&w0, chunk, 8);
    const uint64_t base = w0 & bitmask_u64(p->base_bits);
    const size_t in_chunk = index & (((size_t)1 << p->chunk_len_log2) - 1);
    if (in_chunk == 0) return base;
    const size_t bits = (size_t)p->base_bits + (in_chunk - 1) * p->delta_bits;
    uint64_t w;
    memcpy(&w, chunk + bits / 8, 8);
    return base + ((w >> (bits % 8)) & bitmask_u64(p->delta_bits));
}

/* compress(values) -> bytes + the three parameter bytes; returns the compressed size */
uint64_t qo_ordered_compress(const uint64_t *values, uint64_t n, uint8_t *out, uint64_t cap, uint8_t *params3) {
    const ordered_params p = ordered_find_best(values, (size_t)n);
    bytevec v = {0};
    ordered_compress(&v, values, (size_t)n, &p);
    const uint64_t len = v.len;
    if (out && len <= cap) memcpy(out, v.p, len);
    free(v.p);
    if (params3) {
        params3[0] = p.base_bits;
        params3[1] = p.delta_bits;
        params3[2] = p.chunk_len_log2;
    }
    return len;
}
/* compress_with_parameters for every admissible chunk_len_log2 (the reference test walks try_all) */
uint64_t qo_ordered_compress_with(const uint64_t *values, uint64_t n, uint8_t base_bits, uint8_t delta_bits, uint8_t chunk_len_log2,
                                  uint8_t *out, uint64_t cap) {
    ordered_params p = {n, base_bits, delta_bits, chunk_len_log2};
    bytevec v = {0};
    ordered_compress(&v, values, (size_t)n, &p);
    const uint64_t len = v.len;
    if (out && len <= cap) memcpy(out, v.p, len);
    free(v.p);
    return len;
}
uint64_t qo_ordered_get(const uint8_t *data, uint64_t length, uint8_t base_bits, uint8_t delta_bits, uint8_t chunk_len_log2, uint64_t index) {
    ordered_params p = {length, base_bits, delta_bits, chunk_len_log2};
    return ordered_get(data, &p, (size_t)index);
}

/* ---- serializer.rs: Compressed / CompressedWithVectors ---------------------------------------- */
static size_t varint_put(bytevec *v, uint64_t x) {   /* VarIntWriter::write_varint, unsigned LEB128 */
    size_t n = 0;
    do {
        uint8_t b = (uint8_t)(x & 0x7F);
        x >>= 7;
        if (x) b |= 0x80;
        bv_push(v, &b, 1);
        n++;
    } while (x);
    return n;
}
static size_t next_multiple(size_t x, size_t a) { return a ? (x + a - 1) / a * a : x; }
static void put_u64(uint8_t *dst, uint64_t v) { memcpy(dst, &v, 8); }

/* serialize_graph_links(edges, Compressed | CompressedWithVectors, HnswM{m, m0}) over a graph handed in as the
 * plain GraphLinks arrays (edges[id][level] = neighbors[offsets[idx] .. offsets[idx + 1]], idx as in view.rs
 * offset_idx).  `reindex` fixes back_index (the reference's sort_unstable_by_key leaves the order among equal
 * levels unspecified; any permutation consistent with `reindex` is a valid file).
 * with_vectors != 0: base vector of point i = base_vectors + i * base_size (level 0 only), link vector of
 * point i = link_vectors + i * link_size; sizes must be multiples of the alignments (:33-47).
 * Returns the file size (call with out == NULL to size). */
uint64_t qo_links_serialize_compressed(uint32_t m, uint32_t m0, uint32_t n_points, uint32_t n_levels, const uint32_t *reindex,
                                       const uint64_t *level_offsets, const uint64_t *offsets, const uint32_t *neighbors,
                                       int with_vectors, uint64_t base_size, uint8_t base_align, const uint8_t *base_vectors,
                                       uint64_t link_size, uint8_t link_align, const uint8_t *link_vectors, uint8_t *out, uint64_t cap) {
    bytevec f = {0};
    uint8_t bpu = packed_bits_u32(n_points ? n_points - 1 : 0);
    if (bpu < MIN_BITS_PER_VALUE) bpu = MIN_BITS_PER_VALUE;
    const size_t header_size = with_vectors ? 80 : 64;
    bv_zeros(&f, header_size);                                             /* 1. header placeholder */
    const uint64_t total_offsets_len = (n_levels ? level_offsets[n_levels] : 0) + 1;
    for (uint32_t l = 0; l < n_levels; l++) bv_push(&f, &level_offsets[l], 8);   /* 2. level offsets */
    bv_push(&f, reindex, (size_t)n_points * 4);                            /* 3. reindex */
    if (with_vectors) {                                                    /* 4. neighbors padding */
        const size_t al = base_align > link_align ? base_align : link_align;
        bv_zeros(&f, next_multiple(f.len, al) - f.len);
    }
    uint32_t *back_index = (uint32_t *)malloc(sizeof(uint32_t) * (n_points ? n_points : 1));
    for (uint32_t i = 0; i < n_points; i++) back_index[reindex[i]] = i;
    uint64_t *offs = (uint64_t *)malloc(sizeof(uint64_t) * total_offsets_len);
    size_t n_offs = 0, offset = 0;
    offs[n_offs++] = 0;
    bytevec links_buf = {0};
    size_t max_run = 1;
    for (uint64_t i = 0; i + 1 < total_offsets_len; i++)
        if (offsets[i + 1] - offsets[i] > max_run) max_run = (size_t)(offsets[i + 1] - offsets[i]);
    uint32_t *raw = (uint32_t *)malloc(sizeof(uint32_t) * max_run);
    for (uint32_t level = 0; level < n_levels; level++) {                  /* 5. neighbors */
        const uint64_t count = level_offsets[level + 1] - level_offsets[level];
        const uint32_t level_m = level == 0 ? m0 : m;
        for (uint64_t j = 0; j < count; j++) {
            const uint32_t id = level == 0 ? (uint32_t)j : back_index[j];
            const uint64_t idx = level_offsets[level] + j;
            const size_t n = (size_t)(offsets[idx + 1] - offsets[idx]);
            memcpy(raw, neighbors + offsets[idx], n * 4);
            links_buf.len = 0;
            if (!with_vectors) {
                pack_links(&links_buf, raw, n, bpu, level_m);
                bv_push(&f, links_buf.p, links_buf.len);
                offset += links_buf.len;
            } else {
                if (level == 0) {
                    bv_push(&f, base_vectors + (size_t)id * base_size, base_size);
                    offset += base_size;
                }
                offset += varint_put(&f, n);
                pack_links(&links_buf, raw, n, bpu, level_m);
                bv_push(&f, links_buf.p, links_buf.len);
                offset += links_buf.len;
                size_t pad = next_multiple(offset, link_align) - offset;
                bv_zeros(&f, pad);
                offset += pad;
                for (size_t k = 0; k < n; k++) {                           /* same order as raw_links after pack_links */
                    bv_push(&f, link_vectors + (size_t)raw[k] * link_size, link_size);
                    offset += link_size;
                }
                if (level == 0) {
                    pad = next_multiple(offset, base_align) - offset;
                    bv_zeros(&f, pad);
                    offset += pad;
                }
            }
            offs[n_offs++] = offset;
        }
    }
    const ordered_params p = ordered_find_best(offs, n_offs);              /* 7. offsets */
    ordered_compress(&f, offs, n_offs, &p);
    /* 8. header (header.rs:22-54): LittleU64 fields are byte-aligned, so Parameters (11 bytes) packs tight */
    uint8_t *h = f.p;
    put_u64(h + 0, n_points);
    put_u64(h + 8, with_vectors ? 0xFFFFFFFFFFFFFF02ull : 0xFFFFFFFFFFFFFF01ull);
    put_u64(h + 16, n_levels);
    put_u64(h + 24, offset);
    put_u64(h + 32, p.length);
    h[40] = p.base_bits;
    h[41] = p.delta_bits;
    h[42] = p.chunk_len_log2;
    put_u64(h + 43, m);
    put_u64(h + 51, m0);
    if (with_vectors) {
        put_u64(h + 59, base_size);
        h[67] = base_align;
        put_u64(h + 68, link_size);
        h[76] = link_align;
    }
    const uint64_t len = f.len;
    if (out && len <= cap) memcpy(out, f.p, len);
    free(raw);
    free(links_buf.p);
    free(offs);
    free(back_index);
    free(f.p);
    return len;
}

// links_file.hip — host-side reader of the reference's graph-links files (Plain, Compressed, CompressedWithVectors).
//
// Replaces GraphLinksView::load + links() (lib/segment/src/index/hnsw_index/graph_links/view.rs:110-208, 244-275) for
// the search side: the reference keeps the file mmap'ed and unpacks a node's links on every visit
// (iterate_packed_links, lib/common/common/src/bitpacking_links.rs:90-119; offsets through
// bitpacking_ordered::Reader::decode_chunk, bitpacking_ordered.rs:303-316).  A device walk wants fixed-width u32
// links in HBM, so the file is unpacked ONCE here into the plain arrays qmx_hnsw_create uploads.  No device code in
// this file and no device needed: qmx_graph_links_decode works (and is tested) on a box without a GPU.
//
// Layouts read (all little-endian):
//   Plain                  header.rs:9-20    64 B  {point_count, levels_count, total_neighbors_count, total_offset_count,
//                                                   offsets_padding_bytes, [u8; 24]}
//   Compressed             header.rs:22-37   64 B  {point_count, version = ..FF01, levels_count, total_neighbors_bytes,
//                                                   Parameters{length u64, base_bits, delta_bits, chunk_len_log2} (11 B, packed),
//                                                   m u64 @43, m0 u64 @51, [u8; 5]}
//   CompressedWithVectors  header.rs:39-54   80 B  the same up to m0, then base {size u64 @59, align u8 @67},
//                                                   link {size u64 @68, align u8 @76}, [u8; 3]
//   then level_offsets [levels] u64, reindex [points] u32, (..FF02: padding to max(align)), neighbors, offsets.
//   A packed link list (bitpacking_links.rs:38-82): 5 bits (bits_per_sorted - 8), then min(n, level_m) ascending links as
//   deltas of bits_per_sorted bits, then the remaining links of bits_per_unsorted = max(8, bit_width(point_count - 1))
//   bits each, zero padded to a byte.  In the Compressed format a list's link count is implied by its byte length
//   (both widths are >= 8 bits, the padding < 8); in ..FF02 a LEB128 varint holds it.
#include <string.h>

#include <new>
#include <vector>

#include "common.hpp"

namespace qmx {
namespace {

constexpr uint64_t VERSION_COMPRESSED = 0xFFFFFFFFFFFFFF01ull;
constexpr uint64_t VERSION_COMPRESSED_WITH_VECTORS = 0xFFFFFFFFFFFFFF02ull;
constexpr unsigned MIN_LINK_BITS = 8, LINK_HEADER_BITS = 5;

struct LinksOwner {
    std::vector<uint32_t> reindex, neighbors;
    std::vector<uint64_t> level_offsets, offsets;
};

inline uint64_t load_u64(const uint8_t *p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}
inline unsigned bit_width_u64(uint64_t v) { return v ? 64u - (unsigned)__builtin_clzll(v) : 0u; }
inline uint64_t low_mask(unsigned bits) { return bits >= 64 ? ~0ull : (1ull << bits) - 1ull; }

// `bits` (<= 56) bits starting at bit position `pos` of data[0 .. len); bytes past `len` read as zero
inline uint64_t bits_at(const uint8_t *data, uint64_t len, uint64_t pos, unsigned bits) {
    const uint64_t byte = pos >> 3;
    uint64_t w = 0;
    if (byte + 8 <= len) w = load_u64(data + byte);
    else
        for (uint64_t i = byte; i < len; ++i) w |= (uint64_t)data[i] << (8 * (i - byte));
    return (w >> (pos & 7)) & low_mask(bits);
}

// the sorted offsets array: chunk c = [base : base_bits][delta : delta_bits] x (chunk_len - 1), byte aligned
struct OrderedOffsets {
    const uint8_t *data = nullptr;
    uint64_t size = 0;            // bytes of all chunks, without the 7-byte tail
    uint64_t length = 0;
    unsigned base_bits = 0, delta_bits = 0, chunk_log2 = 0;
    uint64_t chunk_bytes = 0;
    uint64_t get(uint64_t index) const {
        const uint8_t *chunk = data + (index >> chunk_log2) * chunk_bytes;
        const uint64_t avail = size + 7 - (uint64_t)(chunk - data);
        // base_bits can be 64: two reads
        uint64_t base = base_bits <= 56 ? bits_at(chunk, avail, 0, base_bits)
                                        : (bits_at(chunk, avail, 0, 32) | (bits_at(chunk, avail, 32, base_bits - 32) << 32));
        const uint64_t k = index & ((1ull << chunk_log2) - 1);
        if (k == 0) return base;
        return base + bits_at(chunk, avail, (uint64_t)base_bits + (k - 1) * delta_bits, delta_bits);
    }
};

int32_t open_offsets(const uint8_t *params11, const uint8_t *data, uint64_t avail, OrderedOffsets &o) {
    o.length = load_u64(params11);
    o.base_bits = params11[8];
    o.delta_bits = params11[9];
    o.chunk_log2 = params11[10];
    // Parameters::validate (bitpacking_ordered.rs:165-182)
    QMX_REQUIRE(o.base_bits >= 1 && o.base_bits <= 64 && o.delta_bits >= 1 && o.delta_bits <= 56 && o.chunk_log2 <= 7, QMX_ERR_BAD_ARG,
                "links file: invalid offsets parameters (base_bits %u, delta_bits %u, chunk_len_log2 %u)", o.base_bits, o.delta_bits,
                o.chunk_log2);
    o.chunk_bytes = ((uint64_t)o.base_bits + (uint64_t)o.delta_bits * ((1ull << o.chunk_log2) - 1) + 7) / 8;
    const uint64_t chunk_len = 1ull << o.chunk_log2;
    QMX_REQUIRE(o.length <= avail, QMX_ERR_BAD_ARG, "links file: offsets length %llu exceeds the file", (unsigned long long)o.length);
    const uint64_t chunks = (o.length + chunk_len - 1) / chunk_len;
    o.size = chunks * o.chunk_bytes;
    QMX_REQUIRE(o.size + 7 <= avail, QMX_ERR_BAD_ARG, "links file truncated in the compressed offsets (%llu > %llu bytes)",
                (unsigned long long)(o.size + 7), (unsigned long long)avail);
    o.data = data;
    return QMX_OK;
}

// one packed list, `n_links` known (or implied: n_links == UINT64_MAX) -> appended to `out`; returns bytes consumed in *used
int32_t unpack_links(const uint8_t *p, uint64_t len, unsigned bits_unsorted, uint64_t level_m, uint64_t n_links, uint64_t point_count,
                     std::vector<uint32_t> &out, uint64_t *used) {
    *used = 0;
    if (len == 0 || n_links == 0) {
        QMX_REQUIRE(n_links == 0 || n_links == UINT64_MAX, QMX_ERR_BAD_ARG, "links file: %llu links in an empty list",
                    (unsigned long long)n_links);
        return QMX_OK;
    }
    uint64_t pos = 0, n_sorted = 0, n_unsorted = 0;
    unsigned bits_sorted = 0;
    const uint64_t total_bits = len * 8;
    if (level_m != 0) {
        bits_sorted = (unsigned)bits_at(p, len, 0, LINK_HEADER_BITS) + MIN_LINK_BITS;
        pos = LINK_HEADER_BITS;
        if (n_links == UINT64_MAX) {
            const uint64_t fit = (total_bits - pos) / bits_sorted;
            n_sorted = fit < level_m ? fit : level_m;
            n_unsorted = (total_bits - pos - n_sorted * bits_sorted) / bits_unsorted;
        } else {
            n_sorted = n_links < level_m ? n_links : level_m;
            n_unsorted = n_links - n_sorted;
        }
    } else {
        n_unsorted = n_links == UINT64_MAX ? total_bits / bits_unsorted : n_links;
    }
    const uint64_t need = pos + n_sorted * bits_sorted + n_unsorted * bits_unsorted;
    QMX_REQUIRE(need <= total_bits, QMX_ERR_BAD_ARG, "links file: a packed list needs %llu bits of %llu", (unsigned long long)need,
                (unsigned long long)total_bits);
    uint32_t acc = 0;
    for (uint64_t i = 0; i < n_sorted; ++i, pos += bits_sorted) {
        acc += (uint32_t)bits_at(p, len, pos, bits_sorted);
        QMX_REQUIRE(acc < point_count, QMX_ERR_OUT_OF_BOUNDS, "link %u out of range", acc);
        out.push_back(acc);
    }
    for (uint64_t i = 0; i < n_unsorted; ++i, pos += bits_unsorted) {
        const uint32_t v = (uint32_t)bits_at(p, len, pos, bits_unsorted);
        QMX_REQUIRE(v < point_count, QMX_ERR_OUT_OF_BOUNDS, "link %u out of range", v);
        out.push_back(v);
    }
    *used = (pos + 7) / 8;
    return QMX_OK;
}

int32_t decode_plain(const uint8_t *b, uint64_t n_bytes, LinksOwner &o, qmx_graph_links &g) {
    uint64_t hdr[5];
    memcpy(hdr, b, sizeof(hdr));
    const uint64_t point_count = hdr[0], levels_count = hdr[1], total_neighbors = hdr[2], total_offsets = hdr[3], pad = hdr[4];
    QMX_REQUIRE(point_count <= 0xFFFFFFFFull && levels_count <= 64 && (pad == 0 || pad == 4), QMX_ERR_BAD_ARG, "not a plain links header");
    QMX_REQUIRE(total_neighbors <= n_bytes / 4 && total_offsets <= n_bytes / 8 && point_count <= n_bytes / 4, QMX_ERR_BAD_ARG,
                "links header counts exceed the file size");
    const uint64_t off_levels = 64, off_reindex = off_levels + levels_count * 8, off_neigh = off_reindex + point_count * 4,
                   off_offsets = off_neigh + total_neighbors * 4 + pad, end = off_offsets + total_offsets * 8;
    QMX_REQUIRE(end <= n_bytes, QMX_ERR_BAD_ARG, "links file truncated (%llu > %llu)", (unsigned long long)end, (unsigned long long)n_bytes);
    QMX_REQUIRE(total_offsets >= 1, QMX_ERR_BAD_ARG, "empty offsets section");
    o.level_offsets.resize((size_t)levels_count + 1);
    memcpy(o.level_offsets.data(), b + off_levels, (size_t)levels_count * 8);
    o.level_offsets[(size_t)levels_count] = total_offsets - 1;
    o.offsets.resize((size_t)total_offsets);
    memcpy(o.offsets.data(), b + off_offsets, (size_t)total_offsets * 8);
    o.reindex.resize((size_t)point_count);
    memcpy(o.reindex.data(), b + off_reindex, (size_t)point_count * 4);
    o.neighbors.resize((size_t)total_neighbors);
    memcpy(o.neighbors.data(), b + off_neigh, (size_t)total_neighbors * 4);
    for (uint64_t i = 0; i < total_neighbors; ++i)
        QMX_REQUIRE(o.neighbors[i] < point_count, QMX_ERR_OUT_OF_BOUNDS, "link %u out of range", o.neighbors[i]);
    for (uint64_t i = 0; i + 1 < total_offsets; ++i)
        QMX_REQUIRE(o.offsets[i] <= o.offsets[i + 1], QMX_ERR_BAD_ARG, "links file: offsets not ascending at %llu", (unsigned long long)i);
    QMX_REQUIRE(o.offsets[(size_t)total_offsets - 1] <= total_neighbors, QMX_ERR_BAD_ARG, "links file: offsets run past the neighbors section");
    for (uint64_t l = 0; l < levels_count; ++l)
        QMX_REQUIRE(o.level_offsets[l] <= o.level_offsets[l + 1], QMX_ERR_BAD_ARG, "links file: level offsets not ascending");
    QMX_REQUIRE(levels_count == 0 || o.level_offsets[0] == 0, QMX_ERR_BAD_ARG, "links file: level 0 does not start at slot 0");
    for (uint64_t i = 0; i < point_count; ++i)
        QMX_REQUIRE(o.reindex[i] < point_count, QMX_ERR_OUT_OF_BOUNDS, "reindex entry out of range");
    g.format = 0;
    g.m = g.m0 = 0;
    g.n_points = (uint32_t)point_count;
    g.n_levels = (uint32_t)levels_count;
    return QMX_OK;
}

int32_t decode_compressed(const uint8_t *b, uint64_t n_bytes, bool with_vectors, LinksOwner &o, qmx_graph_links &g) {
    const uint64_t header_size = with_vectors ? 80 : 64;
    QMX_REQUIRE(n_bytes >= header_size, QMX_ERR_BAD_ARG, "links file shorter than its %llu-byte header", (unsigned long long)header_size);
    const uint64_t point_count = load_u64(b), levels_count = load_u64(b + 16), neighbors_bytes = load_u64(b + 24);
    const uint64_t m = load_u64(b + 43), m0 = load_u64(b + 51);
    QMX_REQUIRE(point_count <= 0xFFFFFFFFull, QMX_ERR_BAD_ARG, "Too many points in GraphLinks file");
    QMX_REQUIRE(levels_count <= 64 && m <= 0xFFFFFFFFull && m0 <= 0xFFFFFFFFull, QMX_ERR_BAD_ARG, "not a compressed links header");
    uint64_t base_size = 0, base_align = 1, link_size = 0, link_align = 1;
    if (with_vectors) {
        base_size = load_u64(b + 59);
        base_align = b[67];
        link_size = load_u64(b + 68);
        link_align = b[76];
        // Layout::from_size_align: power-of-two alignment; NonZero link size (view.rs:196-198)
        QMX_REQUIRE(base_align && !(base_align & (base_align - 1)) && link_align && !(link_align & (link_align - 1)), QMX_ERR_BAD_ARG,
                    "Invalid vector layout");
        QMX_REQUIRE(link_size != 0, QMX_ERR_BAD_ARG, "Zero link vector size in GraphLinks file");
        QMX_REQUIRE(base_size <= n_bytes && link_size <= n_bytes, QMX_ERR_BAD_ARG, "links header vector sizes exceed the file size");
    }
    QMX_REQUIRE(point_count <= n_bytes / 4 && neighbors_bytes <= n_bytes, QMX_ERR_BAD_ARG, "links header counts exceed the file size");
    const uint64_t off_levels = header_size, off_reindex = off_levels + levels_count * 8;
    uint64_t off_neigh = off_reindex + point_count * 4;
    if (with_vectors) {
        const uint64_t al = base_align > link_align ? base_align : link_align;
        off_neigh = (off_neigh + al - 1) / al * al;
    }
    const uint64_t off_offsets = off_neigh + neighbors_bytes;
    QMX_REQUIRE(off_offsets <= n_bytes, QMX_ERR_BAD_ARG, "links file truncated (%llu > %llu)", (unsigned long long)off_offsets,
                (unsigned long long)n_bytes);
    OrderedOffsets oo;
    QMX_TRY(open_offsets(b + 32, b + off_offsets, n_bytes - off_offsets, oo));
    QMX_REQUIRE(oo.length >= 1, QMX_ERR_BAD_ARG, "Total offset count should be at least 1 in GraphLinks file");
    const uint64_t n_lists = oo.length - 1;

    o.level_offsets.resize((size_t)levels_count + 1);
    memcpy(o.level_offsets.data(), b + off_levels, (size_t)levels_count * 8);
    o.level_offsets[(size_t)levels_count] = n_lists;
    o.reindex.resize((size_t)point_count);
    memcpy(o.reindex.data(), b + off_reindex, (size_t)point_count * 4);
    for (uint64_t l = 0; l < levels_count; ++l)
        QMX_REQUIRE(o.level_offsets[l] <= o.level_offsets[l + 1], QMX_ERR_BAD_ARG, "links file: level offsets not ascending");
    QMX_REQUIRE(levels_count == 0 || o.level_offsets[0] == 0, QMX_ERR_BAD_ARG, "links file: level 0 does not start at slot 0");

    unsigned bits_unsorted = bit_width_u64(point_count ? point_count - 1 : 0);
    if (bits_unsorted < MIN_LINK_BITS) bits_unsorted = MIN_LINK_BITS;
    const uint8_t *nb = b + off_neigh;
    o.offsets.assign((size_t)n_lists + 1, 0);
    o.neighbors.clear();
    o.neighbors.reserve((size_t)(neighbors_bytes / 2));
    uint64_t level = 0;
    uint64_t start = oo.get(0);
    for (uint64_t i = 0; i < n_lists; ++i) {
        while (level + 1 < levels_count && i >= o.level_offsets[level + 1]) ++level;
        const uint64_t level_m = level == 0 ? m0 : m;
        const uint64_t end = oo.get(i + 1);
        QMX_REQUIRE(start <= end && end <= neighbors_bytes, QMX_ERR_BAD_ARG, "links file: list %llu spans bytes %llu..%llu of %llu",
                    (unsigned long long)i, (unsigned long long)start, (unsigned long long)end, (unsigned long long)neighbors_bytes);
        o.offsets[i] = o.neighbors.size();
        uint64_t used = 0;
        if (!with_vectors) {
            QMX_TRY(unpack_links(nb + start, end - start, bits_unsorted, level_m, UINT64_MAX, point_count, o.neighbors, &used));
        } else if (end > start) {
            // [base vector (level 0)] [varint count] [packed links] [pad to link_align] [count link vectors] [pad to base_align (level 0)]
            uint64_t pos = start;
            if (level == 0) pos += base_size;
            uint64_t count = 0;
            unsigned shift = 0;
            bool done = false;
            while (pos < end && shift < 64) {
                const uint8_t byte = nb[pos++];
                count |= (uint64_t)(byte & 0x7F) << shift;
                shift += 7;
                if (!(byte & 0x80)) {
                    done = true;
                    break;
                }
            }
            QMX_REQUIRE(done && pos <= end, QMX_ERR_BAD_ARG, "links file: bad varint in list %llu", (unsigned long long)i);
            QMX_REQUIRE(count <= (end - pos), QMX_ERR_BAD_ARG, "links file: list %llu claims %llu links in %llu bytes", (unsigned long long)i,
                        (unsigned long long)count, (unsigned long long)(end - pos));
            QMX_TRY(unpack_links(nb + pos, end - pos, bits_unsorted, level_m, count, point_count, o.neighbors, &used));
            // the offsets count bytes from the start of the neighbors section, so alignment is relative to it (serializer.rs:146-148)
            uint64_t after = pos + used;
            after = (after + link_align - 1) / link_align * link_align;
            QMX_REQUIRE(count <= (n_bytes / link_size) && after + count * link_size <= end, QMX_ERR_BAD_ARG,
                        "links file: list %llu has no room for its %llu link vectors", (unsigned long long)i, (unsigned long long)count);
        }
        start = end;
    }
    o.offsets[(size_t)n_lists] = o.neighbors.size();
    for (uint64_t i = 0; i < point_count; ++i)
        QMX_REQUIRE(o.reindex[i] < point_count, QMX_ERR_OUT_OF_BOUNDS, "reindex entry out of range");
    g.format = with_vectors ? 2 : 1;
    g.m = (uint32_t)m;
    g.m0 = (uint32_t)m0;
    g.n_points = (uint32_t)point_count;
    g.n_levels = (uint32_t)levels_count;
    return QMX_OK;
}

}  // namespace
}  // namespace qmx

using namespace qmx;

extern "C" {

int32_t qmx_graph_links_decode(const void *bytes, uint64_t n_bytes, qmx_graph_links *out) {
    QMX_REQUIRE(bytes && out, QMX_ERR_BAD_ARG, "NULL argument");
    memset(out, 0, sizeof(*out));
    QMX_REQUIRE(n_bytes >= 64, QMX_ERR_BAD_ARG, "links file shorter than its 64-byte header");
    const uint8_t *b = (const uint8_t *)bytes;
    LinksOwner *o = new (std::nothrow) LinksOwner();
    QMX_REQUIRE(o, QMX_ERR_OUT_OF_MEMORY, "out of host memory");
    qmx_graph_links g;
    memset(&g, 0, sizeof(g));
    int32_t rc;
    try {
        // the compressed headers hold their version where the plain header holds levels_count (graph_links/mod.rs: format detection)
        const uint64_t version = load_u64(b + 8);
        if (version == VERSION_COMPRESSED) rc = decode_compressed(b, n_bytes, false, *o, g);
        else if (version == VERSION_COMPRESSED_WITH_VECTORS) rc = decode_compressed(b, n_bytes, true, *o, g);
        else rc = decode_plain(b, n_bytes, *o, g);
    } catch (const std::bad_alloc &) {
        set_error("out of host memory while decoding the links file");
        rc = QMX_ERR_OUT_OF_MEMORY;
    }
    if (rc != QMX_OK) {
        delete o;
        return rc;
    }
    g.n_offsets = o->offsets.size();
    g.n_neighbors = o->neighbors.size();
    g.reindex = o->reindex.data();
    g.level_offsets = o->level_offsets.data();
    g.offsets = o->offsets.data();
    g.neighbors = o->neighbors.data();
    g.owner = o;
    *out = g;
    return QMX_OK;
}

void qmx_graph_links_free(qmx_graph_links *links) {
    if (!links) return;
    delete static_cast<LinksOwner *>(links->owner);
    memset(links, 0, sizeof(*links));
}

int32_t qmx_hnsw_create_from_file(const void *bytes, uint64_t n_bytes, const qmx_hnsw_desc *desc, qmx_hnsw **out) {
    QMX_REQUIRE(bytes && desc && out, QMX_ERR_BAD_ARG, "NULL argument");
    *out = nullptr;
    qmx_graph_links g;
    QMX_TRY(qmx_graph_links_decode(bytes, n_bytes, &g));
    qmx_hnsw_desc d = *desc;
    int32_t rc = QMX_OK;
    if (g.format != 0) {
        if ((desc->m && desc->m != g.m) || (desc->m0 && desc->m0 != g.m0)) {
            set_error("links file header says m = %u, m0 = %u but the descriptor asks for m = %u, m0 = %u", g.m, g.m0, desc->m, desc->m0);
            rc = QMX_ERR_BAD_ARG;
        }
        d.m = g.m;
        d.m0 = g.m0;
    }
    if (rc == QMX_OK) {
        d.n_points = g.n_points;
        d.n_levels = g.n_levels;
        d.reindex = g.reindex;
        d.level_offsets = g.level_offsets;
        d.offsets = g.offsets;
        d.n_offsets = g.n_offsets;
        d.neighbors = g.neighbors;
        d.n_neighbors = g.n_neighbors;
        rc = qmx_hnsw_create(&d, out);
    }
    qmx_graph_links_free(&g);
    return rc;
}

}  // extern "C"

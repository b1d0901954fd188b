// quant_meta.hip — host-side reader of the quantizers' metadata files ("quantized.meta.json",
// lib/segment/src/vector_storage/quantized/quantized_vectors/config.rs:13).
//
// The reference writes each quantizer's `Metadata` with serde_json (`atomic_save_json`, encoded_vectors_u8.rs:306):
//   SQ int8  encoded_vectors_u8.rs:43-47 (#[serde(untagged)] enum -> the bare struct) + :84-91
//            {"actual_dim", "alpha", "offset", "multiplier", "vector_parameters"}
//   PQ       encoded_vectors_pq.rs:46-51
//            {"centroids": [[f32; dim]; <= 256], "vector_division": [{"start", "end"}...], "vector_parameters"}
//   BQ       encoded_vectors_binary.rs:112-125
//            {"vector_parameters", "encoding"?: "OneBit"|"TwoBits"|"OneAndHalfBits",
//             "query_encoding"?: "SameAsStorage"|"Scalar4bits"|"Scalar8bits", "vector_stats"?: {"elements_stats": [{min,max,mean,stddev}]}}
//   VectorParameters  encoded_vectors.rs:28-39  {"dim", "distance_type": "Cosine"|"Dot"|"L1"|"L2", "invert", "count"?}
// and reads it back in `load`.  This file is that read side for the C-ABI: qmx_quant_meta_parse fills the parameter
// structs qmx_segment_create takes (qmx_sq_params / qmx_pq_params / qmx_bq_params), so a segment directory can be opened
// with qmx_segment_create_from_files without the caller re-implementing serde.  Host only; needs no device.
//
// Numbers: serde_json parses a float literal to f64 and narrows to f32 (`as f32`); the same two steps here (strtod, cast),
// so a value written by the reference (ryu's shortest round-trip text) comes back bit-identical.
#include <errno.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <string>
#include <vector>

#include "common.hpp"

namespace qmx {
namespace {

struct JsonValue {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    bool is_integer = false;     // literal had no fraction / exponent
    uint64_t u = 0;              // valid when is_integer and non-negative
    std::string str;
    std::vector<JsonValue> items;
    std::vector<std::pair<std::string, JsonValue>> fields;
    const JsonValue *get(const char *key) const {
        for (const auto &f : fields)
            if (f.first == key) return &f.second;
        return nullptr;
    }
};

struct JsonParser {
    const char *p, *end;
    const char *err = nullptr;
    int depth = 0;

    void ws() {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
    }
    bool fail(const char *what) {
        if (!err) err = what;
        return false;
    }
    bool literal(const char *s) {
        const size_t n = strlen(s);
        if ((size_t)(end - p) < n || memcmp(p, s, n) != 0) return fail("bad literal");
        p += n;
        return true;
    }
    bool string(std::string &out) {
        if (p >= end || *p != '"') return fail("expected a string");
        ++p;
        out.clear();
        while (p < end && *p != '"') {
            unsigned char c = (unsigned char)*p++;
            if (c < 0x20) return fail("control character in a string");
            if (c != '\\') {
                out.push_back((char)c);
                continue;
            }
            if (p >= end) return fail("unterminated escape");
            const char e = *p++;
            switch (e) {
                case '"': out.push_back('"'); break;
                case '\\': out.push_back('\\'); break;
                case '/': out.push_back('/'); break;
                case 'b': out.push_back('\b'); break;
                case 'f': out.push_back('\f'); break;
                case 'n': out.push_back('\n'); break;
                case 'r': out.push_back('\r'); break;
                case 't': out.push_back('\t'); break;
                case 'u': {
                    if (end - p < 4) return fail("short \\u escape");
                    unsigned cp = 0;
                    for (int i = 0; i < 4; ++i) {
                        const char h = *p++;
                        cp <<= 4;
                        if (h >= '0' && h <= '9') cp |= (unsigned)(h - '0');
                        else if (h >= 'a' && h <= 'f') cp |= (unsigned)(h - 'a' + 10);
                        else if (h >= 'A' && h <= 'F') cp |= (unsigned)(h - 'A' + 10);
                        else return fail("bad \\u escape");
                    }
                    // keys and enum names of these files are ASCII; encode the BMP code point as UTF-8 (surrogates kept as is)
                    if (cp < 0x80) out.push_back((char)cp);
                    else if (cp < 0x800) {
                        out.push_back((char)(0xC0 | (cp >> 6)));
                        out.push_back((char)(0x80 | (cp & 0x3F)));
                    } else {
                        out.push_back((char)(0xE0 | (cp >> 12)));
                        out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
                        out.push_back((char)(0x80 | (cp & 0x3F)));
                    }
                    break;
                }
                default: return fail("bad escape");
            }
        }
        if (p >= end) return fail("unterminated string");
        ++p;
        return true;
    }
    bool number(JsonValue &v) {
        const char *s = p;
        if (p < end && *p == '-') ++p;
        if (p >= end || *p < '0' || *p > '9') return fail("bad number");
        if (*p == '0') ++p;
        else
            while (p < end && *p >= '0' && *p <= '9') ++p;
        bool integer = true;
        if (p < end && *p == '.') {
            integer = false;
            ++p;
            if (p >= end || *p < '0' || *p > '9') return fail("bad fraction");
            while (p < end && *p >= '0' && *p <= '9') ++p;
        }
        if (p < end && (*p == 'e' || *p == 'E')) {
            integer = false;
            ++p;
            if (p < end && (*p == '+' || *p == '-')) ++p;
            if (p >= end || *p < '0' || *p > '9') return fail("bad exponent");
            while (p < end && *p >= '0' && *p <= '9') ++p;
        }
        const std::string text(s, p);
        v.kind = JsonValue::Number;
        v.num = strtod(text.c_str(), nullptr);
        v.is_integer = integer && text[0] != '-';
        if (v.is_integer) {
            errno = 0;
            v.u = strtoull(text.c_str(), nullptr, 10);
            if (errno == ERANGE) v.is_integer = false;
        }
        return true;
    }
    bool value(JsonValue &v) {
        if (++depth > 64) return fail("nesting too deep");
        ws();
        if (p >= end) return fail("unexpected end");
        bool ok = true;
        switch (*p) {
            case '{': {
                v.kind = JsonValue::Object;
                ++p;
                ws();
                if (p < end && *p == '}') {
                    ++p;
                    break;
                }
                while (ok) {
                    ws();
                    std::string key;
                    if (!(ok = string(key))) break;
                    ws();
                    if (p >= end || *p != ':') {
                        ok = fail("expected ':'");
                        break;
                    }
                    ++p;
                    v.fields.emplace_back(std::move(key), JsonValue());
                    if (!(ok = value(v.fields.back().second))) break;
                    ws();
                    if (p < end && *p == ',') {
                        ++p;
                        continue;
                    }
                    if (p < end && *p == '}') {
                        ++p;
                        break;
                    }
                    ok = fail("expected ',' or '}'");
                }
                break;
            }
            case '[': {
                v.kind = JsonValue::Array;
                ++p;
                ws();
                if (p < end && *p == ']') {
                    ++p;
                    break;
                }
                while (ok) {
                    v.items.emplace_back();
                    if (!(ok = value(v.items.back()))) break;
                    ws();
                    if (p < end && *p == ',') {
                        ++p;
                        continue;
                    }
                    if (p < end && *p == ']') {
                        ++p;
                        break;
                    }
                    ok = fail("expected ',' or ']'");
                }
                break;
            }
            case '"':
                v.kind = JsonValue::String;
                ok = string(v.str);
                break;
            case 't':
                v.kind = JsonValue::Bool;
                v.b = true;
                ok = literal("true");
                break;
            case 'f':
                v.kind = JsonValue::Bool;
                v.b = false;
                ok = literal("false");
                break;
            case 'n':
                v.kind = JsonValue::Null;
                ok = literal("null");
                break;
            default: ok = number(v);
        }
        --depth;
        return ok;
    }
};

struct MetaOwner {
    std::vector<float> centroids, mean, stddev, tq_shift, tq_scale;
};

// field accessors with serde's strictness: a missing or mistyped required field is an error
int32_t want_u32(const JsonValue &o, const char *key, uint32_t *out) {
    const JsonValue *v = o.get(key);
    QMX_REQUIRE(v && v->kind == JsonValue::Number && v->is_integer && v->u <= 0xFFFFFFFFull, QMX_ERR_BAD_ARG,
                "metadata: field \"%s\" missing or not an unsigned integer", key);
    *out = (uint32_t)v->u;
    return QMX_OK;
}
int32_t want_f32(const JsonValue &o, const char *key, float *out) {
    const JsonValue *v = o.get(key);
    QMX_REQUIRE(v && v->kind == JsonValue::Number, QMX_ERR_BAD_ARG, "metadata: field \"%s\" missing or not a number", key);
    *out = (float)v->num;   // serde_json: f64 then `as f32`
    return QMX_OK;
}
int32_t want_bool(const JsonValue &o, const char *key, uint8_t *out) {
    const JsonValue *v = o.get(key);
    QMX_REQUIRE(v && v->kind == JsonValue::Bool, QMX_ERR_BAD_ARG, "metadata: field \"%s\" missing or not a bool", key);
    *out = v->b ? 1 : 0;
    return QMX_OK;
}

int32_t vector_parameters(const JsonValue &root, qmx_quant_meta &m) {
    const JsonValue *vp = root.get("vector_parameters");
    QMX_REQUIRE(vp && vp->kind == JsonValue::Object, QMX_ERR_BAD_ARG, "metadata: \"vector_parameters\" missing");
    QMX_TRY(want_u32(*vp, "dim", &m.dim));
    QMX_TRY(want_bool(*vp, "invert", &m.invert));
    const JsonValue *dt = vp->get("distance_type");
    QMX_REQUIRE(dt && dt->kind == JsonValue::String, QMX_ERR_BAD_ARG, "metadata: \"distance_type\" missing");
    // DistanceType (encoded_vectors.rs:12-26); old files say "Dot" for cosine segments too (same scorer for SQ / PQ / BQ)
    if (dt->str == "Cosine") m.distance = QMX_DISTANCE_COSINE;
    else if (dt->str == "Dot") m.distance = QMX_DISTANCE_DOT;
    else if (dt->str == "L1") m.distance = QMX_DISTANCE_MANHATTAN;
    else if (dt->str == "L2") m.distance = QMX_DISTANCE_EUCLID;
    else {
        set_error("metadata: unknown distance_type \"%s\"", dt->str.c_str());
        return QMX_ERR_BAD_ARG;
    }
    const JsonValue *cnt = vp->get("count");     // deprecated_count, optional
    if (cnt && cnt->kind != JsonValue::Null) {
        QMX_REQUIRE(cnt->kind == JsonValue::Number && cnt->is_integer, QMX_ERR_BAD_ARG, "metadata: \"count\" is not an unsigned integer");
        m.deprecated_count = cnt->u;
        m.has_deprecated_count = 1;
    }
    return QMX_OK;
}

int32_t parse_sq(const JsonValue &root, qmx_quant_meta &m) {
    QMX_TRY(want_u32(root, "actual_dim", &m.sq.actual_dim));
    QMX_TRY(want_f32(root, "alpha", &m.sq.alpha));
    QMX_TRY(want_f32(root, "offset", &m.sq.offset));
    QMX_TRY(want_f32(root, "multiplier", &m.sq.multiplier));
    m.sq.invert = m.invert;
    QMX_REQUIRE(m.sq.actual_dim >= m.dim && m.sq.actual_dim % 16 == 0 && m.sq.actual_dim - m.dim < 16, QMX_ERR_BAD_ARG,
                "metadata: actual_dim %u is not dim %u rounded up to 16", m.sq.actual_dim, m.dim);
    return QMX_OK;
}

int32_t parse_pq(const JsonValue &root, qmx_quant_meta &m, MetaOwner &o) {
    const JsonValue *c = root.get("centroids");
    QMX_REQUIRE(c && c->kind == JsonValue::Array && !c->items.empty() && c->items.size() <= 256, QMX_ERR_BAD_ARG,
                "metadata: \"centroids\" missing or not 1..256 rows");
    o.centroids.resize(c->items.size() * (size_t)m.dim);
    for (size_t i = 0; i < c->items.size(); ++i) {
        const JsonValue &row = c->items[i];
        QMX_REQUIRE(row.kind == JsonValue::Array && row.items.size() == m.dim, QMX_ERR_BAD_ARG, "metadata: centroid %zu is not %u numbers", i,
                    m.dim);
        for (uint32_t j = 0; j < m.dim; ++j) {
            QMX_REQUIRE(row.items[j].kind == JsonValue::Number, QMX_ERR_BAD_ARG, "metadata: centroid %zu[%u] is not a number", i, j);
            o.centroids[i * m.dim + j] = (float)row.items[j].num;
        }
    }
    // vector_division = get_vector_division(dim, chunk_size) (encoded_vectors_pq.rs:164-169): consecutive ranges of chunk_size
    const JsonValue *d = root.get("vector_division");
    QMX_REQUIRE(d && d->kind == JsonValue::Array && !d->items.empty(), QMX_ERR_BAD_ARG, "metadata: \"vector_division\" missing");
    uint32_t chunk = 0, at = 0;
    for (size_t i = 0; i < d->items.size(); ++i) {
        uint32_t s = 0, e = 0;
        QMX_REQUIRE(d->items[i].kind == JsonValue::Object, QMX_ERR_BAD_ARG, "metadata: vector_division[%zu] is not a range", i);
        QMX_TRY(want_u32(d->items[i], "start", &s));
        QMX_TRY(want_u32(d->items[i], "end", &e));
        if (i == 0) chunk = e - s;
        const uint32_t expect_end = at + chunk < m.dim ? at + chunk : m.dim;
        QMX_REQUIRE(s == at && e == expect_end && e > s, QMX_ERR_NOT_SUPPORTED,
                    "metadata: vector_division[%zu] = %u..%u is not the uniform division by %u of dim %u", i, s, e, chunk, m.dim);
        at = e;
    }
    QMX_REQUIRE(at == m.dim, QMX_ERR_BAD_ARG, "metadata: vector_division covers %u of %u dimensions", at, m.dim);
    m.pq.chunk_size = chunk;
    m.pq.n_centroids = (uint32_t)c->items.size();
    m.pq.centroids = o.centroids.data();
    m.pq.invert = m.invert;
    m.pq.lut_mfma = 0;
    return QMX_OK;
}

int32_t parse_bq(const JsonValue &root, qmx_quant_meta &m, MetaOwner &o) {
    m.bq.encoding = QMX_BQ_ONE_BIT;          // #[serde(default)]
    m.bq_query_encoding = QMX_BQ_QUERY_SAME_AS_STORAGE;
    if (const JsonValue *e = root.get("encoding")) {
        QMX_REQUIRE(e->kind == JsonValue::String, QMX_ERR_BAD_ARG, "metadata: \"encoding\" is not a string");
        if (e->str == "OneBit") m.bq.encoding = QMX_BQ_ONE_BIT;
        else if (e->str == "TwoBits") m.bq.encoding = QMX_BQ_TWO_BITS;
        else if (e->str == "OneAndHalfBits") m.bq.encoding = QMX_BQ_ONE_AND_HALF_BITS;
        else {
            set_error("metadata: unknown encoding \"%s\"", e->str.c_str());
            return QMX_ERR_BAD_ARG;
        }
    }
    if (const JsonValue *e = root.get("query_encoding")) {
        QMX_REQUIRE(e->kind == JsonValue::String, QMX_ERR_BAD_ARG, "metadata: \"query_encoding\" is not a string");
        if (e->str == "SameAsStorage") m.bq_query_encoding = QMX_BQ_QUERY_SAME_AS_STORAGE;
        else if (e->str == "Scalar4bits") m.bq_query_encoding = QMX_BQ_QUERY_SCALAR_4BITS;
        else if (e->str == "Scalar8bits") m.bq_query_encoding = QMX_BQ_QUERY_SCALAR_8BITS;
        else {
            set_error("metadata: unknown query_encoding \"%s\"", e->str.c_str());
            return QMX_ERR_BAD_ARG;
        }
    }
    const JsonValue *vs = root.get("vector_stats");
    if (vs && vs->kind != JsonValue::Null) {
        QMX_REQUIRE(vs->kind == JsonValue::Object, QMX_ERR_BAD_ARG, "metadata: \"vector_stats\" is not an object");
        const JsonValue *es = vs->get("elements_stats");
        QMX_REQUIRE(es && es->kind == JsonValue::Array && es->items.size() == m.dim, QMX_ERR_BAD_ARG,
                    "metadata: \"elements_stats\" missing or not dim = %u entries", m.dim);
        o.mean.resize(m.dim);
        o.stddev.resize(m.dim);
        for (uint32_t i = 0; i < m.dim; ++i) {
            QMX_REQUIRE(es->items[i].kind == JsonValue::Object, QMX_ERR_BAD_ARG, "metadata: elements_stats[%u] is not an object", i);
            float lo, hi;
            QMX_TRY(want_f32(es->items[i], "min", &lo));
            QMX_TRY(want_f32(es->items[i], "max", &hi));
            QMX_TRY(want_f32(es->items[i], "mean", &o.mean[i]));
            QMX_TRY(want_f32(es->items[i], "stddev", &o.stddev[i]));
        }
        m.bq.mean = o.mean.data();
        m.bq.stddev = o.stddev.data();
    }
    m.bq.query_encoding = m.bq_query_encoding;
    return QMX_OK;
}

// EncodedVectorsTQ Metadata (encoded_vectors_tq.rs:33-46): {"vector_parameters", "bits": "bits4" | "bits2" | "bits1_5" | "bits1",
// "mode": "normal" | "plus", "error_correction": null | {"shift", "scale"}, "rotation"?: "padded" | "unpadded"} (serde rename_all = snake_case)
int32_t parse_tq(const JsonValue &root, qmx_quant_meta &m, MetaOwner &o) {
    const JsonValue *b = root.get("bits");
    QMX_REQUIRE(b && b->kind == JsonValue::String, QMX_ERR_BAD_ARG, "metadata: \"bits\" is missing or not a string");
    if (b->str == "bits4") m.tq.bits = QMX_TQ_BITS4;
    else if (b->str == "bits2") m.tq.bits = QMX_TQ_BITS2;
    else if (b->str == "bits1_5") m.tq.bits = QMX_TQ_BITS1_5;
    else if (b->str == "bits1") m.tq.bits = QMX_TQ_BITS1;
    else {
        set_error("metadata: unknown TQBits \"%s\"", b->str.c_str());
        return QMX_ERR_BAD_ARG;
    }
    const JsonValue *md = root.get("mode");
    QMX_REQUIRE(md && md->kind == JsonValue::String, QMX_ERR_BAD_ARG, "metadata: \"mode\" is missing or not a string");
    if (md->str == "normal") m.tq.plus_mode = 0;
    else if (md->str == "plus") m.tq.plus_mode = 1;      // (the per-coordinate shift / scale arrays travel in qmx_tq_params.ec_shift / ec_scale)
    else {
        set_error("metadata: unknown TQMode \"%s\"", md->str.c_str());
        return QMX_ERR_BAD_ARG;
    }
    m.tq.rotation_unpadded = 0;                           // #[serde(default = "default_rotation")]: Padded
    if (const JsonValue *r = root.get("rotation")) {
        QMX_REQUIRE(r->kind == JsonValue::String, QMX_ERR_BAD_ARG, "metadata: \"rotation\" is not a string");
        if (r->str == "padded") m.tq.rotation_unpadded = 0;
        else if (r->str == "unpadded") m.tq.rotation_unpadded = 1;
        else {
            set_error("metadata: unknown TQRotation \"%s\"", r->str.c_str());
            return QMX_ERR_BAD_ARG;
        }
    }
    m.tq.invert = m.invert;
    // error_correction: Option<ErrorCorrectionMetadata {shift, scale}> (:93-96); lengths are checked against padded_dim by qmx_segment_create's caller
    // contract (new_error_correction_from_metadata :70-91): here they must match each other and be present in Plus mode
    const JsonValue *ec = root.get("error_correction");
    if (ec && ec->kind == JsonValue::Object) {
        const JsonValue *sh = ec->get("shift"), *sc = ec->get("scale");
        QMX_REQUIRE(sh && sc && sh->kind == JsonValue::Array && sc->kind == JsonValue::Array && sh->items.size() == sc->items.size(), QMX_ERR_BAD_ARG,
                    "metadata: error_correction needs \"shift\" and \"scale\" arrays of one length");
        o.tq_shift.resize(sh->items.size());
        o.tq_scale.resize(sc->items.size());
        for (size_t i = 0; i < sh->items.size(); ++i) {
            QMX_REQUIRE(sh->items[i].kind == JsonValue::Number && sc->items[i].kind == JsonValue::Number, QMX_ERR_BAD_ARG,
                        "metadata: error_correction entry %zu is not a number", i);
            o.tq_shift[i] = (float)sh->items[i].num;
            o.tq_scale[i] = (float)sc->items[i].num;
        }
        uint64_t d = m.dim, padded = m.tq.bits == QMX_TQ_BITS1 ? (d + 7) / 8 * 8 : m.tq.bits == QMX_TQ_BITS1_5 ? (d * 3 / 2 + 7) / 8 * 8
                                   : m.tq.bits == QMX_TQ_BITS2 ? (d + 3) / 4 * 4 : (d + 1) / 2 * 2;
        QMX_REQUIRE(o.tq_shift.size() == padded, QMX_ERR_BAD_ARG, "metadata: ErrorCorrection length %zu, expected the padded dim %llu", o.tq_shift.size(),
                    (unsigned long long)padded);
        m.tq.ec_shift = o.tq_shift.data();
        m.tq.ec_scale = o.tq_scale.data();
    } else {
        QMX_REQUIRE(!ec || ec->kind == JsonValue::Null, QMX_ERR_BAD_ARG, "metadata: \"error_correction\" is neither null nor an object");
    }
    QMX_REQUIRE(!m.tq.plus_mode || m.tq.ec_shift, QMX_ERR_BAD_ARG, "metadata: mode \"plus\" without error_correction");
    return QMX_OK;
}

}  // namespace
}  // namespace qmx

using namespace qmx;

extern "C" {

int32_t qmx_quant_meta_parse(uint32_t dtype, const char *json, uint64_t n_bytes, qmx_quant_meta *out) {
    QMX_REQUIRE(json && out, QMX_ERR_BAD_ARG, "NULL argument");
    memset(out, 0, sizeof(*out));
    QMX_REQUIRE(dtype == QMX_DTYPE_SQ_U8 || dtype == QMX_DTYPE_PQ || dtype == QMX_DTYPE_BQ || dtype == QMX_DTYPE_TQ, QMX_ERR_BAD_ARG,
                "dtype %u has no quantizer metadata", dtype);
    int32_t rc = QMX_OK;
    MetaOwner *o = nullptr;
    qmx_quant_meta m;
    memset(&m, 0, sizeof(m));
    m.dtype = dtype;
    try {
        JsonValue root;
        JsonParser ps{json, json + n_bytes};
        bool ok = ps.value(root);
        if (ok) {
            ps.ws();
            if (ps.p != ps.end) ok = ps.fail("trailing characters");
        }
        QMX_REQUIRE(ok, QMX_ERR_BAD_ARG, "metadata: JSON error at byte %llu: %s", (unsigned long long)(ps.p - json), ps.err ? ps.err : "?");
        QMX_REQUIRE(root.kind == JsonValue::Object, QMX_ERR_BAD_ARG, "metadata: the document is not an object");
        o = new MetaOwner();
        rc = vector_parameters(root, m);
        if (rc == QMX_OK) {
            if (dtype == QMX_DTYPE_SQ_U8) rc = parse_sq(root, m);
            else if (dtype == QMX_DTYPE_PQ) rc = parse_pq(root, m, *o);
            else if (dtype == QMX_DTYPE_TQ) rc = parse_tq(root, m, *o);
            else rc = parse_bq(root, m, *o);
        }
    } catch (const std::bad_alloc &) {
        set_error("out of host memory while parsing the metadata");
        rc = QMX_ERR_OUT_OF_MEMORY;
    }
    if (rc != QMX_OK) {
        delete o;
        return rc;
    }
    m.owner = o;
    *out = m;
    return QMX_OK;
}

void qmx_quant_meta_free(qmx_quant_meta *meta) {
    if (!meta) return;
    delete static_cast<MetaOwner *>(meta->owner);
    memset(meta, 0, sizeof(*meta));
}

}  // extern "C"

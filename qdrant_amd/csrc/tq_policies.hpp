// tq_policies.hpp - the lane policies of EncodedVectorsTQ (scan_tq.hip has the reference map): shared by the scan / pair kernels (scan_tq.hip), the
// graph walk (hnsw_tq.hip) and the graph build (hnsw_build_tq.hip), one translation unit each (the build alone instantiates 24 kernels).
#pragma once
#include "hnsw_build.hpp"

namespace qmx {

__device__ __forceinline__ int32_t sdot4(uint32_t a, uint32_t b, int32_t c) { return __builtin_amdgcn_sdot4((int)a, (int)b, c, false); }

// codebook values as signed bytes: c_signed = CODEBOOK_U8 - 128 (query4bit/mod.rs:64-72, query2bit/mod.rs)
//   4 bits: -128 -97 -76 -59 -44 -31 -18 -6 | 6 18 31 44 59 76 97 127
//   2 bits: -128 -38 38 127
constexpr uint32_t TQ4_T0 = 0xC5B49F80u, TQ4_T1 = 0xFAEEE1D4u, TQ4_T2 = 0x2C1F1206u, TQ4_T3 = 0x7F614C3Bu;
constexpr uint32_t TQ2_T = 0x7F26DA80u;

// sel: four 4-bit selectors, one per byte -> the four codebook bytes
__device__ __forceinline__ uint32_t tq4_lookup(uint32_t sel) {
    const uint32_t s = sel & 0x07070707u;
    const uint32_t lo = __builtin_amdgcn_perm(TQ4_T1, TQ4_T0, s);      // selector byte 0..3 -> T0, 4..7 -> T1
    const uint32_t hi = __builtin_amdgcn_perm(TQ4_T3, TQ4_T2, s);
    // byte i of the result = lo's byte i, or hi's where the selector was >= 8: a third byte permute with selector i + 4 * bit 3 (two full-rate
    // instructions for the selector; the 0xFF mask of the bit-select form took a quarter-rate 32-bit multiply: 10 issue slots per lookup against 6)
    const uint32_t pick = ((sel >> 1) & 0x04040404u) | 0x03020100u;
    return __builtin_amdgcn_perm(hi, lo, pick);
}

template <bool L2>
__device__ __forceinline__ float tq_postprocess(float dot, const unsigned char *q_lds, uint32_t rid, const ScanArgs &args) {
    const QueryAux *aux = reinterpret_cast<const QueryAux *>(q_lds + args.aux_off);
    const float sf = args.tq_sf[rid];
    float score;
    if (L2) {
        const float ql = __uint_as_float(aux->pad[0]), l2 = args.tq_l2[rid];
        const float a = ql * ql, b = l2 * l2, c = (2.0f * dot) * sf;
        score = (a + b) - c;
    } else {
        score = dot * sf;
    }
    return args.tq_invert ? -score : score;
}

template <bool L2>
struct RowTQ4 {
    static constexpr bool TEMPORAL_ROWS = true;
    static constexpr int NACC = 2;      // sum low * c, sum high * c
    static constexpr int NRAUX = 0;
    static constexpr int R16 = 2;
    static constexpr int QPIECES = 4;
    static constexpr int MAX_VALU_QT = 4;       // 4 queries and more: the int8 matrix cores (scan_sq_mfma.hip TqOps)
    typedef uint32_t acc_t;
    static __device__ __forceinline__ void row_aux(acc_t (&)[1], const uint4 &) {}
    static __device__ __forceinline__ void mac(acc_t (&)[NACC], const uint4 &, const uint4 &) {}
    struct dec_t { uint32_t ce[4], co[4]; };    // codebook bytes of the even / odd dims of the piece
    static __device__ __forceinline__ void decode(const uint4 &v, dec_t &d) {
        const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            d.ce[w] = tq4_lookup(vv[w] & 0x0F0F0F0Fu);
            d.co[w] = tq4_lookup((vv[w] >> 4) & 0x0F0F0F0Fu);
        }
    }
    static __device__ __forceinline__ void mac_decoded(acc_t (&a)[NACC], const uint4 (&q)[4], const dec_t &d) {
        const uint32_t le[4] = {q[0].x, q[0].y, q[0].z, q[0].w}, lo[4] = {q[1].x, q[1].y, q[1].z, q[1].w};
        const uint32_t he[4] = {q[2].x, q[2].y, q[2].z, q[2].w}, ho[4] = {q[3].x, q[3].y, q[3].z, q[3].w};
        int32_t al = (int32_t)a[0], ah = (int32_t)a[1];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            al = sdot4(le[w], d.ce[w], al);
            al = sdot4(lo[w], d.co[w], al);
            ah = sdot4(he[w], d.ce[w], ah);
            ah = sdot4(ho[w], d.co[w], ah);
        }
        a[0] = (uint32_t)al;
        a[1] = (uint32_t)ah;
    }
    static __device__ __forceinline__ void mac_pieces(acc_t (&a)[NACC], const uint4 (&q)[4], const uint4 &v) {
        dec_t d;
        decode(v, d);
        mac_decoded(a, q, d);
    }
    static __device__ __forceinline__ float finish(acc_t (&a)[NACC], acc_t (&)[1], const unsigned char *q_lds, const unsigned char *, uint32_t rid,
                                                   const ScanArgs &args) {
        const int64_t low = (int64_t)(int32_t)reduce8_u32(a[0]), high = (int64_t)(int32_t)reduce8_u32(a[1]);
        const int64_t s = low + 128 * high;                                   // = dot_raw - bias_correction
        const QueryAux *aux = reinterpret_cast<const QueryAux *>(q_lds + args.aux_off);
        const float raw = aux->f0 * (float)s;
        return tq_postprocess<L2>(raw + __uint_as_float(aux->pad[3]), q_lds, rid, args);   // + query.ec_correction (0.0 without TQ+)
    }
};

template <bool L2>
struct RowTQ2 {
    static constexpr bool TEMPORAL_ROWS = true;
    static constexpr int NACC = 2;
    static constexpr int NRAUX = 0;
    static constexpr int R16 = 2;
    static constexpr int QPIECES = 8;
    static constexpr int MAX_VALU_QT = 4;
    typedef uint32_t acc_t;
    static __device__ __forceinline__ void row_aux(acc_t (&)[1], const uint4 &) {}
    static __device__ __forceinline__ void mac(acc_t (&)[NACC], const uint4 &, const uint4 &) {}
    struct dec_t { uint32_t c[4][4]; };         // codebook bytes of the dims = j mod 4, per dword
    static __device__ __forceinline__ void decode(const uint4 &v, dec_t &d) {
        const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int w = 0; w < 4; ++w) d.c[j][w] = __builtin_amdgcn_perm(0u, TQ2_T, (vv[w] >> (2 * j)) & 0x03030303u);
    }
    static __device__ __forceinline__ void mac_decoded(acc_t (&a)[NACC], const uint4 (&q)[8], const dec_t &d) {
        int32_t al = (int32_t)a[0], ah = (int32_t)a[1];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t ql[4] = {q[j].x, q[j].y, q[j].z, q[j].w}, qh[4] = {q[4 + j].x, q[4 + j].y, q[4 + j].z, q[4 + j].w};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                al = sdot4(ql[w], d.c[j][w], al);
                ah = sdot4(qh[w], d.c[j][w], ah);
            }
        }
        a[0] = (uint32_t)al;
        a[1] = (uint32_t)ah;
    }
    static __device__ __forceinline__ void mac_pieces(acc_t (&a)[NACC], const uint4 (&q)[8], const uint4 &v) {
        dec_t d;
        decode(v, d);
        mac_decoded(a, q, d);
    }
    static __device__ __forceinline__ float finish(acc_t (&a)[NACC], acc_t (&)[1], const unsigned char *q_lds, const unsigned char *, uint32_t rid,
                                                   const ScanArgs &args) {
        const int64_t low = (int64_t)(int32_t)reduce8_u32(a[0]), high = (int64_t)(int32_t)reduce8_u32(a[1]);
        const int64_t s = low + 128 * high;
        const QueryAux *aux = reinterpret_cast<const QueryAux *>(q_lds + args.aux_off);
        const float raw = aux->f0 * (float)s;
        return tq_postprocess<L2>(raw + __uint_as_float(aux->pad[3]), q_lds, rid, args);   // + query.ec_correction (0.0 without TQ+)
    }
};

template <bool L2, int PLANES = 8>       // PLANES = BITS of Query1bitSimd<BITS>: 8, or 16 under TQ+ (Bits1Wide)
struct RowTQ1 {
    static constexpr bool TEMPORAL_ROWS = true;
    static constexpr int NACC = 1;
    static constexpr int NRAUX = 0;
    static constexpr int R16 = PLANES == 8 ? 2 : 1;
    static constexpr int QPIECES = PLANES;
    static constexpr int MAX_VALU_QT = 4;
    typedef uint32_t acc_t;
    static __device__ __forceinline__ void row_aux(acc_t (&)[1], const uint4 &) {}
    static __device__ __forceinline__ void mac(acc_t (&)[NACC], const uint4 &, const uint4 &) {}
    static __device__ __forceinline__ void mac_pieces(acc_t (&a)[NACC], const uint4 (&q)[PLANES], const uint4 &v) {
        int32_t s = (int32_t)a[0];
#pragma unroll
        for (int k = 0; k < PLANES; ++k) {
            const int32_t c = __popc(q[k].x & v.x) + __popc(q[k].y & v.y) + __popc(q[k].z & v.z) + __popc(q[k].w & v.w);
            s += k == PLANES - 1 ? -(c << k) : (c << k);                    // w_b = 2^b, the sign plane -2^(BITS - 1)
        }
        a[0] = (uint32_t)s;
    }
    static __device__ __forceinline__ float finish(acc_t (&a)[NACC], acc_t (&)[1], const unsigned char *q_lds, const unsigned char *, uint32_t rid,
                                                   const ScanArgs &args) {
        const int64_t v_dot_q = (int64_t)(int32_t)reduce8_u32(a[0]);
        const QueryAux *aux = reinterpret_cast<const QueryAux *>(q_lds + args.aux_off);
        const int64_t sum_q = (int64_t)(((uint64_t)aux->pad[2] << 32) | aux->pad[1]);
        const int64_t signed_dot = 2 * v_dot_q - sum_q;
        const float raw = aux->f0 * (float)signed_dot;
        return tq_postprocess<L2>(raw + __uint_as_float(aux->pad[3]), q_lds, rid, args);   // + query.ec_correction (0.0 without TQ+)
    }
};

template <class L>
static int32_t dispatch_tq(const L &l, const ScanArgs &a) {
    const bool l2 = a.tq_l2 != nullptr;
    switch (a.tq_bits) {
        case 4: return l2 ? l.template row<RowTQ4<true>>(a) : l.template row<RowTQ4<false>>(a);
        case 2: return l2 ? l.template row<RowTQ2<true>>(a) : l.template row<RowTQ2<false>>(a);
        case 1:
            if (a.tq_planes == 16) return l2 ? l.template row<RowTQ1<true, 16>>(a) : l.template row<RowTQ1<false, 16>>(a);
            return l2 ? l.template row<RowTQ1<true>>(a) : l.template row<RowTQ1<false>>(a);
    }
    set_error("TurboQuant: %u bits per value not supported", a.tq_bits);
    return QMX_ERR_NOT_SUPPORTED;
}

// per bit width: scan_tq{4,2,1}.hip
int32_t launch_scan_tq4(hipStream_t st, int qt, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid_out);
int32_t launch_scan_tq2(hipStream_t st, int qt, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid_out);
int32_t launch_scan_tq1(hipStream_t st, int qt, ScanMode mode, const ScanArgs &a, int num_cus, uint32_t *grid_out);
int32_t launch_pairs_tq4(hipStream_t st, const ScanArgs &a, const PairSel &sel, uint64_t n_items, int num_cus);
int32_t launch_pairs_tq2(hipStream_t st, const ScanArgs &a, const PairSel &sel, uint64_t n_items, int num_cus);
int32_t launch_pairs_tq1(hipStream_t st, const ScanArgs &a, const PairSel &sel, uint64_t n_items, int num_cus);

}  // namespace qmx
